"""CPU restatement of the reference's two association loops -- TEST INFRASTRUCTURE ONLY.

    SfM::find2D3DMatches     SfMToyLib/SfM.cpp:471-528
    SfM::mergeNewPointCloud  SfMToyLib/SfM.cpp:530-629   (thresholds SfM.cpp:50-51)

Both are integer / index work on the reference's containers, so parity with the HIP path is BIT-EXACT: same
entries, same order.  The loops are restated on plain Python containers that mirror the reference's:

    cloud            list of (xyz float32[3], views) with views = {view index: feature index}; a std::map, i.e. iterated in
                     ascending view index
    match matrix     dict {(left, right): [(queryIdx, trainIdx, distance), ...]} for left <= right (the reference only ever
                     indexes mFeatureMatchMatrix[smaller][larger], SfM.cpp:489-490,555-558); list order is significant
                     ("first match wins", SfM.cpp:493-512,566-578)

Pure-Python loops: meant for small cases (seconds).  Allowed importers: tests/ only.

parity pinned?  The reference has no test and no golden output for either function (SfMUnitTests.cpp covers only the
camera model, triangulation and homography paths), and the reference cannot be built here (OpenCV, Boost absent):
"parity unpinned" in the sense of the task -- the restatement is checked against hand-derived expected results for the
tie / duplicate / iterate-while-inserting cases (tests/test_oracle_association.py).
"""
import numpy as np

MERGE_CLOUD_POINT_MIN_MATCH_DISTANCE = np.float32(0.01)     # SfM.cpp:50
MERGE_CLOUD_FEATURE_MIN_MATCH_DISTANCE = np.float32(20.0)   # SfM.cpp:51


def find_2d3d_matches(n_views, done_views, cloud, match_matrix):
    """SfM.cpp:471-528.  Returns {view: [(cloud point index, feature index in that view), ...]} for every view that is
    not done, entries in cloud order.  (The reference stores features[view].points[idx] and cloudPoint.p; the indices
    carry the same information and keep the comparison exact.)"""
    done = set(int(v) for v in done_views)
    out = {}
    for view in range(n_views):                                   # :475
        if view in done:                                          # :476-478
            continue
        found = []
        for i, (_, views) in enumerate(cloud):                    # :483
            hit = None
            for ov in sorted(views):                              # :487, std::map order
                of = views[ov]
                orig_is_left = ov < view                          # :497 (ov == view counts as 'right')
                left, right = (ov, view) if orig_is_left else (view, ov)       # :494-495
                for (q, t, _) in match_matrix.get((left, right), ()):          # :498
                    cand = -1
                    if orig_is_left:
                        if q == of:
                            cand = t                              # :500-502
                    elif t == of:
                        cand = q                                  # :504-506
                    if cand >= 0:                                 # :508: a negative index does not stop the scan
                        hit = cand
                        break
                if hit is not None:                               # :518-520
                    break
            if hit is not None:
                found.append((i, int(hit)))
        out[view] = found                                         # :524
    return out


def cv_norm_diff(existing_xyz, new_xyz):
    """cv::norm(existingPoint.p - newPoint) for cv::Point3f (SfM.cpp:544): the difference is formed in float, the norm is
    sqrt((double)x*x + (double)y*y + (double)z*z)  [OpenCV core/types.hpp, norm(Point3_<_Tp>)]."""
    d = (np.asarray(existing_xyz, np.float32) - np.asarray(new_xyz, np.float32)).astype(np.float32)
    d = d.astype(np.float64)
    return float(np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]))


def merge_new_point_cloud(cloud, new_cloud, match_matrix,
                          point_dist=MERGE_CLOUD_POINT_MIN_MATCH_DISTANCE, feature_dist=MERGE_CLOUD_FEATURE_MIN_MATCH_DISTANCE):
    """SfM.cpp:530-629.  `cloud` is modified in place exactly like mReconstructionCloud (views of existing points grow,
    new points are appended); returns (new_points, merged_points, merge_matches) where merge_matches lists the
    (left, right, position in that pair's list) pushed to mergeMatchMatrix (:571), in push order."""
    new_points = merged_points = 0
    merge_matches = []
    pd, fd = float(np.float32(point_dist)), float(np.float32(feature_dist))
    for (new_xyz, new_views) in new_cloud:                        # :538
        found_any_views = False                                   # :541
        found_3d = False                                          # :542
        for (ex_xyz, ex_views) in cloud:                          # :543 -- includes points appended earlier in this call
            if cv_norm_diff(ex_xyz, new_xyz) < pd:                # :544
                found_3d = True
                for nv in sorted(new_views):                      # :549
                    nf = new_views[nv]
                    # :553 iterates existingPoint.originatingViews WHILE :582 inserts into it: a std::map iterator keeps
                    # walking in key order and visits a key inserted behind it if that key is larger than the current one
                    cur = None
                    while True:
                        keys = sorted(k for k in ex_views if cur is None or k > cur)
                        if not keys:
                            break
                        ev = cur = keys[0]
                        ef = ex_views[ev]
                        new_is_left = nv < ev                     # :559
                        left, lf, right, rf = (nv, nf, ev, ef) if new_is_left else (ev, ef, nv, nf)     # :560-563
                        hit = False
                        for pos, (q, t, dist) in enumerate(match_matrix.get((left, right), ())):        # :566
                            if q == lf and t == rf and float(np.float32(dist)) < fd:                      # :567-569
                                merge_matches.append((left, right, pos))                                  # :571
                                hit = True
                                break
                        if hit:
                            ex_views[nv] = nf                     # :582 (insert or overwrite)
                            found_any_views = True
            if found_any_views:                                   # :590-593
                merged_points += 1
                break
        if not found_any_views and not found_3d:                  # :596-600
            cloud.append((np.asarray(new_xyz, np.float32).copy(), dict(new_views)))
            new_points += 1
    return new_points, merged_points, merge_matches


def radius_candidates(existing_xyz, new_xyz, point_dist=MERGE_CLOUD_POINT_MIN_MATCH_DISTANCE):
    """What the HIP radius join must return: for new point k every index j < n_exist + k of the concatenated sequence
    [existing ..., new 0 .. k-1] with cv::norm(seq[j] - new[k]) < point_dist, ascending.  numpy, vectorised per new point."""
    ex = np.asarray(existing_xyz, np.float32).reshape(-1, 3)
    nw = np.asarray(new_xyz, np.float32).reshape(-1, 3)
    seq = np.concatenate([ex, nw]).astype(np.float32)
    pd = np.float64(np.float32(point_dist))
    ptr, idx = [0], []
    for k in range(len(nw)):
        lim = len(ex) + k
        d = (seq[:lim] - nw[k]).astype(np.float32).astype(np.float64)
        with np.errstate(invalid="ignore", over="ignore"):
            nrm = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2])
            hit = np.flatnonzero(nrm < pd)
        idx.extend(int(j) for j in hit)
        ptr.append(len(idx))
    return np.asarray(ptr, np.int64), np.asarray(idx, np.int32)
