// shim_harness.cpp -- flat-array entry point used by tests/test_gpu_shim.py to drive the C++ shim:
// builds the reference's containers (PointCloud / vector<Matx34f> / Intrinsics / vector<Features>),
// calls sfmtoylib::SfMBundleAdjustmentUtils::adjustBundle() and copies the containers back.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include "SfMBundleAdjustmentUtils.h"

extern "C" __attribute__((visibility("default")))
void sfmba_shim_adjust_bundle(int n_views, float* poses /*[n_views][12]*/, float* K /*[9]*/, int n_pts, float* points /*[n_pts][3]*/,
                              const int64_t* view_ptr, const int32_t* view_idx, const int32_t* feat_idx,
                              const int64_t* feat_ptr, const float* feat_xy) {
    using namespace sfmtoylib;
    std::vector<Pose> cams((size_t)n_views);
    for (int v = 0; v < n_views; ++v) for (int e = 0; e < 12; ++e) cams[v].val[e] = poses[12 * v + e];
    Intrinsics intr;
    intr.K = cv::Mat(3, 3);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) intr.K.at<float>(r, c) = K[3 * r + c];
    std::vector<Features> feats((size_t)n_views);
    for (int v = 0; v < n_views; ++v)
        for (int64_t f = feat_ptr[v]; f < feat_ptr[v + 1]; ++f) feats[v].points.push_back(cv::Point2f(feat_xy[2 * f], feat_xy[2 * f + 1]));
    PointCloud cloud((size_t)n_pts);
    for (int i = 0; i < n_pts; ++i) {
        cloud[i].p = cv::Point3f(points[3 * i], points[3 * i + 1], points[3 * i + 2]);
        for (int64_t q = view_ptr[i]; q < view_ptr[i + 1]; ++q) cloud[i].originatingViews[view_idx[q]] = feat_idx[q];
    }
    const auto t0 = std::chrono::steady_clock::now();
    SfMBundleAdjustmentUtils::adjustBundle(cloud, cams, intr, feats);
    if (std::getenv("SFMBA_SHIM_TIMING"))
        std::fprintf(stderr, "[sfmba shim] adjustBundle() wall time %.3f ms\n",
                     1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    for (int v = 0; v < n_views; ++v) for (int e = 0; e < 12; ++e) poses[12 * v + e] = cams[v].val[e];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) K[3 * r + c] = intr.K.at<float>(r, c);
    for (int i = 0; i < n_pts; ++i) { points[3 * i] = cloud[i].p.x; points[3 * i + 1] = cloud[i].p.y; points[3 * i + 2] = cloud[i].p.z; }
}
