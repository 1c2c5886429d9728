// SfMAssociation.h -- the two association loops of sfmtoylib::SfM (SfMToyLib/SfM.cpp:471-528, 530-629) as free-standing
// functions over the members they read and write, backed by the MI355X joins of include/sfmba.h
// (sfmba_find_2d3d_matches, sfmba_merge_candidates).  In the reference both are private member functions; a maintainer
// replaces each body by one call:
//
//   SfM::Images2D3DMatches SfM::find2D3DMatches() {
//       return SfMAssociation::find2D3DMatches(mImages.size(), mDoneViews, mReconstructionCloud, mFeatureMatchMatrix, mImageFeatures);
//   }
//   void SfM::mergeNewPointCloud(const PointCloud& cloud) {
//       SfMAssociation::mergeNewPointCloud(mReconstructionCloud, cloud, mFeatureMatchMatrix);
//   }
//
// Results are identical to the reference loops: same entries, same order, same mutations of the cloud.
#pragma once
#include <map>
#include <set>
#include <vector>

#include "SfMCommon.h"

namespace sfmtoylib {

typedef std::vector<std::vector<Matching> > MatchMatrix;          // SfM.h:50
struct Image2D3DMatch {                                           // SfMCommon.h:71-74
    Points2f points2D;
    Points3f points3D;
};
typedef std::map<int, Image2D3DMatch> Images2D3DMatches;          // SfM.h:52

class SfMAssociation {
public:
    /**
     * For every view that is not done: the 2D features of that view that correspond to 3D points of the cloud
     * (SfM::find2D3DMatches, SfM.cpp:471-528).  Every not-done view gets an entry, possibly empty.
     * On a device error the result is empty and a line is written to stderr (the reference has no error path here).
     */
    static Images2D3DMatches find2D3DMatches(
            size_t                       numImages,
            const std::set<int>&         doneViews,
            const PointCloud&            reconstructionCloud,
            const MatchMatrix&           featureMatchMatrix,
            const std::vector<Features>& imageFeatures);

    /**
     * Merge `cloud` into `reconstructionCloud` (SfM::mergeNewPointCloud, SfM.cpp:530-629): a new point close to an existing
     * one whose 2D features are confirmed by the match matrix adds its views to that point; a new point close to nothing is
     * appended; anything else is dropped.  mergeMatchMatrix (optional) receives the matches the reference collects for
     * its debug display (SfM.cpp:571); newPoints / mergedPoints the two counters it prints (SfM.cpp:626-628).
     * Returns false on a device error (reconstructionCloud untouched).
     */
    static bool mergeNewPointCloud(
            PointCloud&        reconstructionCloud,
            const PointCloud&  cloud,
            const MatchMatrix& featureMatchMatrix,
            MatchMatrix*       mergeMatchMatrix = nullptr,
            size_t*            newPoints        = nullptr,
            size_t*            mergedPoints     = nullptr);
};

}  // namespace sfmtoylib
