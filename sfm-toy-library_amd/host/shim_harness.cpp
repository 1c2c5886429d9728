// shim_harness.cpp -- flat-array entry point used by tests/test_gpu_shim.py to drive the C++ shim:
// builds the reference's containers (PointCloud / vector<Matx34f> / Intrinsics / vector<Features>),
// calls sfmtoylib::SfMBundleAdjustmentUtils::adjustBundle() and copies the containers back.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include "SfMBundleAdjustmentUtils.h"
#include "SfMStereoUtilities.h"
#include "SfMAssociation.h"
#include "SfMExport.h"

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

namespace {
using namespace sfmtoylib;
PointCloud buildCloud(int n, const float* xyz, const int64_t* view_ptr, const int32_t* view_idx, const int32_t* feat_idx) {
    PointCloud cloud((size_t)n);
    for (int i = 0; i < n; ++i) {
        cloud[i].p = cv::Point3f(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
        for (int64_t q = view_ptr[i]; q < view_ptr[i + 1]; ++q) cloud[i].originatingViews[view_idx[q]] = feat_idx[q];
    }
    return cloud;
}
MatchMatrix buildMatchMatrix(int n_views, int n_pairs, const int32_t* left, const int32_t* right, const int64_t* ptr, const int32_t* query,
                             const int32_t* train, const float* dist) {
    MatchMatrix mm((size_t)n_views, std::vector<Matching>((size_t)n_views));
    for (int p = 0; p < n_pairs; ++p)
        for (int64_t e = ptr[p]; e < ptr[p + 1]; ++e) mm[left[p]][right[p]].push_back(cv::DMatch(query[e], train[e], dist ? dist[e] : 0.0f));
    return mm;
}
}  // namespace

// Flat-array driver of sfmtoylib::SfMAssociation::find2D3DMatches (tests/test_gpu_association.py).  Returns the total number of
// 2D-3D pairs (entries beyond cap are not written), -1 if a not-done view has no entry in the result.
extern "C" __attribute__((visibility("default")))
int64_t sfmba_shim_find_2d3d(int n_views, const unsigned char* done, int n_pt, const float* xyz, const int64_t* view_ptr, const int32_t* view_idx,
                             const int32_t* feat_idx, int n_pairs, const int32_t* left, const int32_t* right, const int64_t* pair_ptr,
                             const int32_t* query, const int32_t* train, const int64_t* feat_ptr, const float* feat_xy,
                             int64_t* out_ptr /*[n_views+1]*/, float* out_2d /*[cap][2]*/, float* out_3d /*[cap][3]*/, int64_t cap) {
    using namespace sfmtoylib;
    const PointCloud cloud = buildCloud(n_pt, xyz, view_ptr, view_idx, feat_idx);
    const MatchMatrix mm = buildMatchMatrix(n_views, n_pairs, left, right, pair_ptr, query, train, nullptr);
    std::vector<Features> feats((size_t)n_views);
    for (int v = 0; v < n_views; ++v)
        for (int64_t f = feat_ptr[v]; f < feat_ptr[v + 1]; ++f) feats[v].points.push_back(cv::Point2f(feat_xy[2 * f], feat_xy[2 * f + 1]));
    std::set<int> doneViews;
    for (int v = 0; v < n_views; ++v) if (done[v]) doneViews.insert(v);
    const Images2D3DMatches res = SfMAssociation::find2D3DMatches((size_t)n_views, doneViews, cloud, mm, feats);
    int64_t n = 0;
    for (int v = 0; v < n_views; ++v) {
        out_ptr[v] = n;
        const auto it = res.find(v);
        if (done[v]) { if (it != res.end()) return -1; continue; }
        if (it == res.end()) return -1;
        for (size_t e = 0; e < it->second.points2D.size(); ++e, ++n) {
            if (n >= cap) continue;
            out_2d[2 * n] = it->second.points2D[e].x; out_2d[2 * n + 1] = it->second.points2D[e].y;
            out_3d[3 * n] = it->second.points3D[e].x; out_3d[3 * n + 1] = it->second.points3D[e].y; out_3d[3 * n + 2] = it->second.points3D[e].z;
        }
    }
    out_ptr[n_views] = n;
    return n;
}

// Flat-array driver of sfmtoylib::SfMAssociation::mergeNewPointCloud.  The merged reconstruction cloud comes back flattened
// (out_n points, views CSR); counts[0..1] = new / merged points; merge_pairs = (left, right, query, train) of the matches pushed to
// the merge match matrix, row-major over the matrix.  Returns 0, -1 on device failure, -2 if an output capacity is too small.
extern "C" __attribute__((visibility("default")))
int sfmba_shim_merge(int n_views, int n_exist, const float* ex_xyz, const int64_t* ex_view_ptr, const int32_t* ex_view_idx, const int32_t* ex_feat_idx,
                     int n_new, const float* nw_xyz, const int64_t* nw_view_ptr, const int32_t* nw_view_idx, const int32_t* nw_feat_idx,
                     int n_pairs, const int32_t* left, const int32_t* right, const int64_t* pair_ptr, const int32_t* query, const int32_t* train,
                     const float* dist, int cap_pts, int64_t cap_views, int* out_n, float* out_xyz, int64_t* out_view_ptr, int32_t* out_view_idx,
                     int32_t* out_feat_idx, int64_t* counts, int64_t cap_merge, int32_t* merge_pairs /*[cap_merge][4]*/, int64_t* n_merge) {
    using namespace sfmtoylib;
    PointCloud recon = buildCloud(n_exist, ex_xyz, ex_view_ptr, ex_view_idx, ex_feat_idx);
    const PointCloud fresh = buildCloud(n_new, nw_xyz, nw_view_ptr, nw_view_idx, nw_feat_idx);
    const MatchMatrix mm = buildMatchMatrix(n_views, n_pairs, left, right, pair_ptr, query, train, dist);
    MatchMatrix merged;
    size_t np = 0, nm = 0;
    if (!SfMAssociation::mergeNewPointCloud(recon, fresh, mm, &merged, &np, &nm)) return -1;
    counts[0] = (int64_t)np; counts[1] = (int64_t)nm;
    if ((int)recon.size() > cap_pts) return -2;
    *out_n = (int)recon.size();
    int64_t o = 0;
    for (size_t i = 0; i < recon.size(); ++i) {
        out_xyz[3 * i] = recon[i].p.x; out_xyz[3 * i + 1] = recon[i].p.y; out_xyz[3 * i + 2] = recon[i].p.z;
        out_view_ptr[i] = o;
        for (const auto& kv : recon[i].originatingViews) { if (o >= cap_views) return -2; out_view_idx[o] = kv.first; out_feat_idx[o] = kv.second; ++o; }
    }
    out_view_ptr[recon.size()] = o;
    int64_t m = 0;
    for (size_t l = 0; l < merged.size(); ++l)
        for (size_t r = 0; r < merged[l].size(); ++r)
            for (const cv::DMatch& d : merged[l][r]) {
                if (m < cap_merge) { merge_pairs[4 * m] = (int32_t)l; merge_pairs[4 * m + 1] = (int32_t)r; merge_pairs[4 * m + 2] = d.queryIdx; merge_pairs[4 * m + 3] = d.trainIdx; }
                ++m;
            }
    *n_merge = m;
    return 0;
}

// Flat-array driver of sfmtoylib::SfMExport::saveCloudAndCamerasToPLY (tests/test_ply_export.py).  images: n_views images of
// img_rows x img_cols BGR bytes, concatenated.
extern "C" __attribute__((visibility("default")))
int sfmba_shim_save_ply(const char* prefix, int n_views, const float* poses /*[n_views][12]*/, int n_pt, const float* xyz, const int64_t* view_ptr,
                        const int32_t* view_idx, const int32_t* feat_idx, const int64_t* feat_ptr, const float* feat_xy, int img_rows, int img_cols,
                        const unsigned char* images) {
    using namespace sfmtoylib;
    const PointCloud cloud = buildCloud(n_pt, xyz, view_ptr, view_idx, feat_idx);
    std::vector<cv::Matx34f> cams((size_t)n_views);
    for (int v = 0; v < n_views; ++v) for (int e = 0; e < 12; ++e) cams[v].val[e] = poses[12 * v + e];
    std::vector<Features> feats((size_t)n_views);
    std::vector<ImageBGR> imgs((size_t)n_views);
    for (int v = 0; v < n_views; ++v) {
        for (int64_t f = feat_ptr[v]; f < feat_ptr[v + 1]; ++f) feats[v].points.push_back(cv::Point2f(feat_xy[2 * f], feat_xy[2 * f + 1]));
#ifndef SFMBA_HAVE_OPENCV
        imgs[v].rows = img_rows; imgs[v].cols = img_cols;
        imgs[v].data.assign(images + (size_t)v * img_rows * img_cols * 3, images + (size_t)(v + 1) * img_rows * img_cols * 3);
#endif
    }
    return SfMExport::saveCloudAndCamerasToPLY(prefix, cloud, cams, feats, imgs) ? 0 : -1;
}
