// SfMExport.h -- SfM::saveCloudAndCamerasToPLY (SfMToyLib/SfM.cpp:631-711) as a free-standing function over the members it reads:
// the on-disk format on the far side of the bundle-adjustment path.  Byte-identical output.
// In the reference it is a member function; the replacement body is one call:
//
//   void SfM::saveCloudAndCamerasToPLY(const std::string& prefix) {
//       SfMExport::saveCloudAndCamerasToPLY(prefix, mReconstructionCloud, mCameraPoses, mImageFeatures, mImages);
//   }
#pragma once
#include <string>
#include <vector>

#include "SfMCommon.h"

namespace sfmtoylib {

#ifdef SFMBA_HAVE_OPENCV
typedef cv::Mat ImageBGR;                      // CV_8UC3, as loaded by cv::imread (SfM.cpp:120)
#else
// stand-in for a CV_8UC3 cv::Mat: rows x cols pixels, 3 bytes each in B, G, R order, row-major
struct ImageBGR {
    int rows = 0, cols = 0;
    std::vector<unsigned char> data;
};
#endif

class SfMExport {
public:
    /**
     * Writes <prefix>_points.ply (one vertex per cloud point, coloured by the pixel under the point's first originating view's
     * feature) and <prefix>_cameras.ply (four vertices and three axis edges per camera pose).  Returns false if a file could
     * not be written (the reference does not check).
     */
    static bool saveCloudAndCamerasToPLY(
            const std::string&              prefix,
            const PointCloud&               reconstructionCloud,
            const std::vector<cv::Matx34f>& cameraPoses,
            const std::vector<Features>&    imageFeatures,
            const std::vector<ImageBGR>&    images);
};

}  // namespace sfmtoylib
