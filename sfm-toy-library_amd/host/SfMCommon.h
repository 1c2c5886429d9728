// SfMCommon.h -- boundary data model of the reference (SfMToyLib/SfMCommon.h:55-59,76-88,96,99):
// the types adjustBundle() takes by reference.  Same names, same members, same namespace.
#pragma once
#include <map>
#include <vector>

#include "cv_compat.h"

namespace sfmtoylib {

struct Intrinsics {
    cv::Mat K;
    cv::Mat Kinv;
    cv::Mat distortion;
};

typedef std::vector<cv::KeyPoint> Keypoints;
typedef std::vector<cv::Point2f>  Points2f;
typedef std::vector<cv::Point3f>  Points3f;

struct Features {
    Keypoints keyPoints;
    Points2f  points;
    cv::Mat   descriptors;
};

struct Point3DInMap {
    // 3D point.
    cv::Point3f p;
    // A mapping from image index to 2D point index in that image's list of features.
    std::map<int, int> originatingViews;
};

typedef std::vector<Point3DInMap> PointCloud;
typedef std::vector<cv::DMatch>   Matching;      // SfMCommon.h:95

struct ImagePair {                                // SfMCommon.h:61-63
    size_t left, right;
};
typedef cv::Matx34f Pose;

}  // namespace sfmtoylib
