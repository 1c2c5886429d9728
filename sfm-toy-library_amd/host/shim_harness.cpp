// shim_harness.cpp -- flat-array entry point used by tests/test_gpu_shim.py to drive the C++ shim:
// builds the reference's containers (PointCloud / vector<Matx34f> / Intrinsics / vector<Features>),
// calls sfmtoylib::SfMBundleAdjustmentUtils::adjustBundle() and copies the containers back.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include "SfMBundleAdjustmentUtils.h"
#include "SfMStereoUtilities.h"

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

// Flat-array driver of sfmtoylib::SfMStereoUtilities::triangulateViews (tests/test_gpu_triangulate.py): builds Features /
// Matching, calls the reference-signature function and flattens the resulting PointCloud.  Returns the number of points
// written (<= cap), or -1 when the call reported failure.
extern "C" __attribute__((visibility("default")))
int sfmba_shim_triangulate_views(const float* K /*[9]*/, int left_view, int right_view, int n_left, const float* left_xy, int n_right,
                                 const float* right_xy, int n_match, const int32_t* query_idx, const int32_t* train_idx,
                                 const float* P_left /*[12]*/, const float* P_right /*[12]*/, int cap, float* points3d /*[cap][3]*/,
                                 int32_t* left_ref /*[cap]*/, int32_t* right_ref /*[cap]*/) {
    using namespace sfmtoylib;
    Intrinsics intr;
    intr.K = cv::Mat(3, 3);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) intr.K.at<float>(r, c) = K[3 * r + c];
    Features fl, fr;
    for (int i = 0; i < n_left; ++i) { cv::KeyPoint kp; kp.pt = cv::Point2f(left_xy[2 * i], left_xy[2 * i + 1]); fl.keyPoints.push_back(kp); fl.points.push_back(kp.pt); }
    for (int i = 0; i < n_right; ++i) { cv::KeyPoint kp; kp.pt = cv::Point2f(right_xy[2 * i], right_xy[2 * i + 1]); fr.keyPoints.push_back(kp); fr.points.push_back(kp.pt); }
    Matching matches;
    for (int i = 0; i < n_match; ++i) matches.push_back(cv::DMatch(query_idx[i], train_idx[i], 0.0f));
    cv::Matx34f Pl, Pr;
    for (int e = 0; e < 12; ++e) { Pl.val[e] = P_left[e]; Pr.val[e] = P_right[e]; }
    PointCloud cloud;
    ImagePair pair; pair.left = (size_t)left_view; pair.right = (size_t)right_view;
    if (!SfMStereoUtilities::triangulateViews(intr, pair, matches, fl, fr, Pl, Pr, cloud)) return -1;
    int n = 0;
    for (const Point3DInMap& p : cloud) {
        if (n >= cap) break;
        points3d[3 * n] = p.p.x; points3d[3 * n + 1] = p.p.y; points3d[3 * n + 2] = p.p.z;
        left_ref[n] = p.originatingViews.at(left_view);
        right_ref[n] = p.originatingViews.at(right_view);
        ++n;
    }
    return n;
}
