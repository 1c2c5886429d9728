// SfMAssociation.cpp -- host side of the association joins: flattens the reference's containers, calls the C ABI
// (include/sfmba.h), and rebuilds the reference's results.  See SfMAssociation.h.
#include "SfMAssociation.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <unordered_map>

#include "../../include/sfmba.h"

namespace sfmtoylib {

namespace {

const float MERGE_CLOUD_POINT_MIN_MATCH_DISTANCE   = 0.01f;     // SfM.cpp:50
const float MERGE_CLOUD_FEATURE_MIN_MATCH_DISTANCE = 20.0f;     // SfM.cpp:51

struct FlatCloudViews {
    std::vector<int64_t> ptr;
    std::vector<int32_t> view, feat;
};

FlatCloudViews flattenViews(const PointCloud& cloud) {
    FlatCloudViews f;
    f.ptr.assign(cloud.size() + 1, 0);
    for (size_t i = 0; i < cloud.size(); ++i) f.ptr[i + 1] = f.ptr[i] + (int64_t)cloud[i].originatingViews.size();
    f.view.resize((size_t)f.ptr.back());
    f.feat.resize((size_t)f.ptr.back());
    for (size_t i = 0; i < cloud.size(); ++i) {
        size_t o = (size_t)f.ptr[i];
        for (const auto& kv : cloud[i].originatingViews) { f.view[o] = kv.first; f.feat[o] = kv.second; ++o; }   // ascending view
    }
    return f;
}

struct FlatMatches {
    std::vector<int32_t> left, right, query, train;
    std::vector<int64_t> ptr;
};

// upper triangle (left <= right) of the match matrix, non-empty lists only
FlatMatches flattenMatches(const MatchMatrix& mm, size_t numImages) {
    FlatMatches f;
    f.ptr.push_back(0);
    for (size_t l = 0; l < mm.size() && l < numImages; ++l)
        for (size_t r = l; r < mm[l].size() && r < numImages; ++r) {
            const Matching& m = mm[l][r];
            if (m.empty()) continue;
            f.left.push_back((int32_t)l); f.right.push_back((int32_t)r);
            for (const cv::DMatch& d : m) { f.query.push_back(d.queryIdx); f.train.push_back(d.trainIdx); }
            f.ptr.push_back((int64_t)f.query.size());
        }
    return f;
}

}  // namespace

Images2D3DMatches SfMAssociation::find2D3DMatches(size_t numImages, const std::set<int>& doneViews, const PointCloud& cloud,
                                                  const MatchMatrix& featureMatchMatrix, const std::vector<Features>& imageFeatures) {
    Images2D3DMatches matches;
    const int n_views = (int)numImages;
    std::vector<unsigned char> done((size_t)n_views, 0);
    for (int v : doneViews) if (v >= 0 && v < n_views) done[(size_t)v] = 1;
    const FlatCloudViews cv_ = flattenViews(cloud);
    const FlatMatches fm = flattenMatches(featureMatchMatrix, numImages);
    std::vector<int64_t> out_ptr((size_t)n_views + 1, 0);
    std::vector<int32_t> out_point, out_feat;
    int64_t total = 0, cap = (int64_t)cloud.size();       // one new view's worth; grown on demand
    for (int attempt = 0; attempt < 2; ++attempt) {
        out_point.resize((size_t)cap); out_feat.resize((size_t)cap);
        const int rc = sfmba_find_2d3d_matches(0, n_views, done.data(), (int)cloud.size(), cv_.ptr.data(), cv_.view.data(), cv_.feat.data(),
                                               (int)fm.left.size(), fm.left.data(), fm.right.data(), fm.ptr.data(), fm.query.data(), fm.train.data(),
                                               out_ptr.data(), out_point.data(), out_feat.data(), cap, &total);
        if (rc == SFMBA_OK) break;
        if (rc == SFMBA_ERR_CAPACITY && attempt == 0) { cap = total; continue; }
        std::fprintf(stderr, "find2D3DMatches failed (sfmba rc=%d: %s)\n", rc, sfmba_last_error());
        return matches;
    }
    for (int v = 0; v < n_views; ++v) {
        if (done[(size_t)v]) continue;
        Image2D3DMatch m;                                  // SfM.cpp:480,524: an entry for every not-done view
        const Features& f = imageFeatures[(size_t)v];
        m.points2D.reserve((size_t)(out_ptr[v + 1] - out_ptr[v]));
        m.points3D.reserve((size_t)(out_ptr[v + 1] - out_ptr[v]));
        for (int64_t e = out_ptr[v]; e < out_ptr[v + 1]; ++e) {
            m.points2D.push_back(f.points[(size_t)out_feat[(size_t)e]]);              // SfM.cpp:511
            m.points3D.push_back(cloud[(size_t)out_point[(size_t)e]].p);              // SfM.cpp:512
        }
        matches[v] = m;
    }
    return matches;
}

namespace {

// "is there a match (query, train) with distance below the threshold in this pair's list, and which is the first"
// (SfM.cpp:566-578), answered from an index built once per consulted pair instead of a scan per question.
class PairIndex {
public:
    explicit PairIndex(const Matching& m) {
        for (size_t pos = 0; pos < m.size(); ++pos) {
            if (!(m[pos].distance < MERGE_CLOUD_FEATURE_MIN_MATCH_DISTANCE)) continue;
            const uint64_t key = ((uint64_t)(uint32_t)m[pos].queryIdx << 32) | (uint32_t)m[pos].trainIdx;
            first_.emplace(key, pos);                      // emplace keeps the FIRST position of a duplicate key
        }
    }
    long find(int query, int train) const {
        const auto it = first_.find(((uint64_t)(uint32_t)query << 32) | (uint32_t)train);
        return it == first_.end() ? -1 : (long)it->second;
    }
private:
    std::unordered_map<uint64_t, size_t> first_;
};

}  // namespace

bool SfMAssociation::mergeNewPointCloud(PointCloud& recon, const PointCloud& cloud, const MatchMatrix& featureMatchMatrix,
                                        MatchMatrix* mergeMatchMatrix, size_t* newPointsOut, size_t* mergedPointsOut) {
    const size_t n_exist = recon.size(), n_new = cloud.size();
    size_t newPoints = 0, mergedPoints = 0;
    if (newPointsOut) *newPointsOut = 0;
    if (mergedPointsOut) *mergedPointsOut = 0;
    if (mergeMatchMatrix) {                                // SfM.cpp:532-533: numImages x numImages
        mergeMatchMatrix->resize(featureMatchMatrix.size());
        for (auto& row : *mergeMatchMatrix) row.resize(featureMatchMatrix.size());
    }
    if (n_new == 0) return true;

    // ---- the O(n^2) distance tests: on the device ----
    std::vector<float> ex(3 * n_exist), nw(3 * n_new);
    for (size_t i = 0; i < n_exist; ++i) { ex[3 * i] = recon[i].p.x; ex[3 * i + 1] = recon[i].p.y; ex[3 * i + 2] = recon[i].p.z; }
    for (size_t k = 0; k < n_new; ++k) { nw[3 * k] = cloud[k].p.x; nw[3 * k + 1] = cloud[k].p.y; nw[3 * k + 2] = cloud[k].p.z; }
    std::vector<int64_t> cand_ptr(n_new + 1, 0);
    std::vector<int32_t> cand;
    int64_t total = 0, cap = (int64_t)(4 * n_new + 1024);
    for (int attempt = 0; attempt < 2; ++attempt) {
        cand.resize((size_t)cap);
        const int rc = sfmba_merge_candidates(0, (int)n_exist, ex.data(), (int)n_new, nw.data(), MERGE_CLOUD_POINT_MIN_MATCH_DISTANCE,
                                              cand_ptr.data(), cand.data(), cap, &total);
        if (rc == SFMBA_OK) break;
        if (rc == SFMBA_ERR_CAPACITY && attempt == 0) { cap = total; continue; }
        std::fprintf(stderr, "mergeNewPointCloud failed (sfmba rc=%d: %s)\n", rc, sfmba_last_error());
        return false;
    }

    // ---- the sequential remainder (SfM.cpp:545-600), over the candidates only ----
    std::map<std::pair<int, int>, PairIndex> indices;      // built lazily per consulted view pair
    auto pairIndex = [&](int l, int r) -> const PairIndex* {
        if (l < 0 || r < 0 || (size_t)l >= featureMatchMatrix.size() || (size_t)r >= featureMatchMatrix[(size_t)l].size()) return nullptr;
        const Matching& m = featureMatchMatrix[(size_t)l][(size_t)r];
        if (m.empty()) return nullptr;
        auto it = indices.find(std::make_pair(l, r));
        if (it == indices.end()) it = indices.emplace(std::make_pair(l, r), PairIndex(m)).first;
        return &it->second;
    };
    std::vector<long> appendedAt(n_new, -1);               // position in recon of new point k, if it was appended
    for (size_t k = 0; k < n_new; ++k) {
        const Point3DInMap& p = cloud[k];
        bool anyViewConfirmed = false, anyClosePoint = false;
        for (int64_t c = cand_ptr[k]; c < cand_ptr[k + 1] && !anyViewConfirmed; ++c) {
            const size_t j = (size_t)cand[(size_t)c];
            long at = (long)j;
            if (j >= n_exist) { at = appendedAt[j - n_exist]; if (at < 0) continue; }      // that earlier new point never joined the cloud
            Point3DInMap& existing = recon[(size_t)at];
            anyClosePoint = true;                                                          // SfM.cpp:546
            for (const auto& newKv : p.originatingViews) {
                // the existing point's view map grows while it is walked (SfM.cpp:553 / :582); std::map iterators stay valid
                // and run into keys inserted ahead of them, exactly as in the reference
                for (auto ex_it = existing.originatingViews.begin(); ex_it != existing.originatingViews.end(); ++ex_it) {
                    const bool newIsLeft = newKv.first < ex_it->first;
                    const int l  = newIsLeft ? newKv.first  : ex_it->first,  r  = newIsLeft ? ex_it->first  : newKv.first;
                    const int lf = newIsLeft ? newKv.second : ex_it->second, rf = newIsLeft ? ex_it->second : newKv.second;
                    const PairIndex* idx = pairIndex(l, r);
                    const long pos = idx ? idx->find(lf, rf) : -1;
                    if (pos < 0) continue;
                    if (mergeMatchMatrix) (*mergeMatchMatrix)[(size_t)l][(size_t)r].push_back(featureMatchMatrix[(size_t)l][(size_t)r][(size_t)pos]);
                    existing.originatingViews[newKv.first] = newKv.second;                 // SfM.cpp:582
                    anyViewConfirmed = true;
                }
            }
        }
        if (anyViewConfirmed) { ++mergedPoints; continue; }                                // SfM.cpp:590-593
        if (!anyClosePoint) {                                                              // SfM.cpp:596-600
            appendedAt[k] = (long)recon.size();
            recon.push_back(p);
            ++newPoints;
        }
    }
    if (newPointsOut) *newPointsOut = newPoints;
    if (mergedPointsOut) *mergedPointsOut = mergedPoints;
    return true;
}

}  // namespace sfmtoylib
