// triangulate.hip -- two-view DLT triangulation + reprojection filter (gfx950), SURVEY 8(f) row 2.
//
// Device counterpart of SfMStereoUtilities::triangulateViews for ALIGNED matches
// (SfMToyLib/SfMStereoUtilities.cpp:120-206): per match
//   :145-149  undistortPoints with no distortion         x_n = (u - cx)/fx, y_n = (v - cy)/fy            (-> float)
//   :151-152  cv::triangulatePoints [OpenCV-upstream]    A = [x P3 - P1; y P3 - P2] of both views, 4x4, fp64;
//                                                        X_h = right singular vector of the smallest singular value (-> float)
//   :154-155  convertPointsFromHomogeneous               X = X_h.xyz / X_h.w                               (float)
//   :157-169  projectPoints (Rodrigues(R) == R)          u = fx (R X + t)_x / (R X + t)_z + cx, fp64      (-> float)
//   :183-190  keep iff both reprojection errors <= 10 px (MIN_REPROJECTION_ERROR, :42)
// One lane per match, everything in registers: the 4x4 SVD is a one-sided (Hestenes) Jacobi iteration on the columns of A
// with V accumulated -- 6 sweeps of the 6 column pairs, no LDS, no divergence beyond the rotation skip.  The pass is
// HBM-trivial (16 B in, 13 B out per match); it exists so that the step in front of bundle adjustment need not leave the GPU.
#include "ba_kernels.h"

namespace sfmba {

namespace {

struct TriCams {
    float K[9];
    float Pl[12];
    float Pr[12];
};

__device__ __forceinline__ void jacobi_pair(double A[4][4], double V[4][4], int p, int q) {
    double app = 0.0, aqq = 0.0, apq = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) { app += A[r][p] * A[r][p]; aqq += A[r][q] * A[r][q]; apq += A[r][p] * A[r][q]; }
    if (fabs(apq) <= 1e-300 || fabs(apq) <= 1e-17 * sqrt(app * aqq)) return;
    const double zeta = (aqq - app) / (2.0 * apq);
    const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double ap = A[r][p], aq = A[r][q];
        A[r][p] = c * ap - s * aq; A[r][q] = s * ap + c * aq;
        const double vp = V[r][p], vq = V[r][q];
        V[r][p] = c * vp - s * vq; V[r][q] = s * vp + c * vq;
    }
}

__device__ __forceinline__ float2 project_px(const float* P, const float* K, const float X[3]) {
    const double x = (double)P[0] * X[0] + (double)P[1] * X[1] + (double)P[2] * X[2] + (double)P[3];
    const double y = (double)P[4] * X[0] + (double)P[5] * X[1] + (double)P[6] * X[2] + (double)P[7];
    const double z = (double)P[8] * X[0] + (double)P[9] * X[1] + (double)P[10] * X[2] + (double)P[11];
    return make_float2((float)((double)K[0] * x / z + (double)K[2]), (float)((double)K[4] * y / z + (double)K[5]));
}

__global__ __launch_bounds__(256) void k_triangulate(long long n, const float2* __restrict__ left, const float2* __restrict__ right,
                                                     TriCams cams, float max_err, float* __restrict__ points3d,
                                                     unsigned char* __restrict__ keep, float* __restrict__ err) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float2 l = left[i], r = right[i];
    const double fx = cams.K[0], fy = cams.K[4], cx = cams.K[2], cy = cams.K[5];
    const double xl = (double)(float)(((double)l.x - cx) / fx), yl = (double)(float)(((double)l.y - cy) / fy);
    const double xr = (double)(float)(((double)r.x - cx) / fx), yr = (double)(float)(((double)r.y - cy) / fy);
    double A[4][4], V[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        A[0][c] = xl * (double)cams.Pl[8 + c] - (double)cams.Pl[c];
        A[1][c] = yl * (double)cams.Pl[8 + c] - (double)cams.Pl[4 + c];
        A[2][c] = xr * (double)cams.Pr[8 + c] - (double)cams.Pr[c];
        A[3][c] = yr * (double)cams.Pr[8 + c] - (double)cams.Pr[4 + c];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) V[rr][c] = (rr == c) ? 1.0 : 0.0;
    }
#pragma unroll 1
    for (int sweep = 0; sweep < 8; ++sweep) {
        jacobi_pair(A, V, 0, 1); jacobi_pair(A, V, 0, 2); jacobi_pair(A, V, 0, 3);
        jacobi_pair(A, V, 1, 2); jacobi_pair(A, V, 1, 3); jacobi_pair(A, V, 2, 3);
    }
    // column of the smallest singular value (branch-free selection)
    double best = 0.0, v[4] = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        double nn = 0.0;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) nn += A[rr][c] * A[rr][c];
        const bool take = (c == 0) || (nn < best);
        best = take ? nn : best;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) v[rr] = take ? V[rr][c] : v[rr];
    }
    const float h[4] = { (float)v[0], (float)v[1], (float)v[2], (float)v[3] };
    const float scale = h[3] != 0.0f ? 1.0f / h[3] : 1.0f;
    const float X[3] = { h[0] * scale, h[1] * scale, h[2] * scale };
    const float2 pl = project_px(cams.Pl, cams.K, X), pr = project_px(cams.Pr, cams.K, X);
    const double el = sqrt((double)(pl.x - l.x) * (double)(pl.x - l.x) + (double)(pl.y - l.y) * (double)(pl.y - l.y));
    const double er = sqrt((double)(pr.x - r.x) * (double)(pr.x - r.x) + (double)(pr.y - r.y) * (double)(pr.y - r.y));
    points3d[3 * i] = X[0]; points3d[3 * i + 1] = X[1]; points3d[3 * i + 2] = X[2];
    keep[i] = (el > (double)max_err || er > (double)max_err) ? 0 : 1;
    if (err) { err[2 * i] = (float)el; err[2 * i + 1] = (float)er; }
}

}  // namespace

void launch_triangulate(hipStream_t s, long long n, const float* d_left, const float* d_right, const float K[9], const float Pl[12],
                        const float Pr[12], float max_err, float* d_points3d, unsigned char* d_keep, float* d_err) {
    if (n <= 0) return;
    TriCams cams;
    for (int e = 0; e < 9; ++e) cams.K[e] = K[e];
    for (int e = 0; e < 12; ++e) { cams.Pl[e] = Pl[e]; cams.Pr[e] = Pr[e]; }
    hipLaunchKernelGGL(k_triangulate, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, reinterpret_cast<const float2*>(d_left),
                       reinterpret_cast<const float2*>(d_right), cams, max_err, d_points3d, d_keep, d_err);
}

}  // namespace sfmba
