// SfMStereoUtilities.cpp -- drop-in for SfMStereoUtilities::triangulateViews (SfMToyLib/SfMStereoUtilities.cpp:120-206).
//
// Marshalling mirrors the reference: matches are aligned through queryIdx / trainIdx with back references to the original
// feature indices (GetAlignedPointsFromMatch, SfMCommon.cpp:63-87; the 2D points are the key points' pt, :85-86), the
// triangulation + 10 px reprojection filter runs on the GPU (sfmba_triangulate), and every surviving point is appended to
// pointCloud with originatingViews[left] / [right] = those back references (:192-203).  Points are appended in match order.
#include "SfMStereoUtilities.h"

#include <iostream>
#include <vector>

#include "../../include/sfmba.h"

namespace sfmtoylib {

bool SfMStereoUtilities::triangulateViews(
        const Intrinsics&  intrinsics,
        const ImagePair    imagePair,
        const Matching&    matches,
        const Features&    featuresLeft,
        const Features&    featuresRight,
        const cv::Matx34f& Pleft,
        const cv::Matx34f& Pright,
        PointCloud&        pointCloud) {
    const size_t n = matches.size();
    std::vector<float> left(2 * n), right(2 * n);
    std::vector<int> leftBackReference(n), rightBackReference(n);
    for (size_t i = 0; i < n; i++) {
        const cv::Point2f& pl = featuresLeft.keyPoints[matches[i].queryIdx].pt;
        const cv::Point2f& pr = featuresRight.keyPoints[matches[i].trainIdx].pt;
        left[2 * i] = pl.x;  left[2 * i + 1] = pl.y;
        right[2 * i] = pr.x; right[2 * i + 1] = pr.y;
        leftBackReference[i] = matches[i].queryIdx;
        rightBackReference[i] = matches[i].trainIdx;
    }
    float K[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) K[3 * r + c] = intrinsics.K.at<float>(r, c);

    std::vector<float> points3d(3 * n);
    std::vector<unsigned char> keep(n);
    const float MIN_REPROJECTION_ERROR = 10.0f;     // SfMStereoUtilities.cpp:42
    const int rc = sfmba_triangulate(0, (int64_t)n, left.data(), right.data(), K, Pleft.val, Pright.val, MIN_REPROJECTION_ERROR,
                                     points3d.data(), keep.data(), nullptr);
    if (rc != SFMBA_OK) {
        std::cerr << "triangulateViews failed. (sfmba rc=" << rc << ": " << sfmba_last_error() << ")" << std::endl;
        return false;
    }
    for (size_t i = 0; i < n; i++) {
        if (!keep[i]) continue;
        Point3DInMap p;
        p.p = cv::Point3f(points3d[3 * i], points3d[3 * i + 1], points3d[3 * i + 2]);
        p.originatingViews[(int)imagePair.left]  = leftBackReference[i];
        p.originatingViews[(int)imagePair.right] = rightBackReference[i];
        pointCloud.push_back(p);
    }
    return true;
}

}  // namespace sfmtoylib
