// SfMStereoUtilities.h -- the triangulation entry point of the reference with its own signature
// (SfMToyLib/SfMStereoUtilities.h:72-91), backed by the MI355X kernel (include/sfmba.h: sfmba_triangulate).
// Only triangulateViews is provided: the other members of the reference class (homography inliers, essential-matrix pose,
// PnP) stay on the reference's OpenCV path (out of scope).
#pragma once
#include "SfMCommon.h"

namespace sfmtoylib {

class SfMStereoUtilities {
public:
    /**
     * Triangulate (recover 3D locations) from point matching.
     * @return true on success (false: no HIP device / device error; pointCloud untouched).
     */
    static bool triangulateViews(
            const Intrinsics&  intrinsics,
            const ImagePair    imagePair,
            const Matching&    matches,
            const Features&    leftFeatures,
            const Features&    rightFeatures,
            const cv::Matx34f& Pleft,
            const cv::Matx34f& Pright,
            PointCloud&        pointCloud);
};

}  // namespace sfmtoylib
