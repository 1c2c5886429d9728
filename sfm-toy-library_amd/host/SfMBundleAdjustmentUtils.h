// SfMBundleAdjustmentUtils.h -- same declaration as the reference (SfMToyLib/SfMBundleAdjustmentUtils.h:35-50).
#pragma once
#include "SfMCommon.h"

namespace sfmtoylib {

class SfMBundleAdjustmentUtils {
public:
    /**
     * Global bundle adjustment of all registered cameras, all points and the shared focal length.
     * In-out arguments; left bit-identical unless the solver reports CONVERGENCE (reference BA.cpp:182-185).
     */
    static void adjustBundle(
            PointCloud&                     pointCloud,
            std::vector<cv::Matx34f>&       cameraPoses,
            Intrinsics&                     intrinsics,
            const std::vector<Features>&    image2dFeatures
            );
};

} /* namespace sfmtoylib */
