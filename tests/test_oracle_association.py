"""CPU tests of oracle/association_oracle.py (restatement of SfM::find2D3DMatches / SfM::mergeNewPointCloud,
SfM.cpp:471-629) against HAND-DERIVED results for the cases the loops decide by ORDER: first match wins, duplicates,
negative indices, views inserted while the view map is being iterated.  The reference holds no test for either function."""
import numpy as np

from oracle import association_oracle as ao


def P(x, y, z, views):
    return (np.array([x, y, z], np.float32), dict(views))


def test_find_first_match_in_list_order_wins():
    # view 2 is new; the cloud point was seen as feature 5 of view 0.  Two matches of pair (0,2) carry query 5: the FIRST one
    # in list order decides (SfM.cpp:498-512), not the best distance.
    cloud = [P(0, 0, 1, {0: 5})]
    mm = {(0, 2): [(4, 9, 1.0), (5, 7, 50.0), (5, 3, 1.0)]}
    assert ao.find_2d3d_matches(3, {0, 1}, cloud, mm) == {2: [(0, 7)]}


def test_find_first_originating_view_with_a_match_wins_and_direction_flips():
    # new view 1 sits BETWEEN the originating views 0 and 2: (0,1) is searched by queryIdx, (1,2) by trainIdx (SfM.cpp:497-507)
    cloud = [P(0, 0, 1, {0: 5, 2: 8}), P(1, 0, 1, {2: 8})]
    mm = {(0, 1): [(6, 1, 0.0)],                 # feature 5 of view 0 unmatched -> falls through to view 2
          (1, 2): [(11, 7, 0.0), (12, 8, 0.0), (13, 8, 0.0)]}
    assert ao.find_2d3d_matches(3, {0, 2}, cloud, mm) == {1: [(0, 12), (1, 12)]}


def test_find_negative_index_does_not_stop_the_scan_and_done_views_are_skipped():
    cloud = [P(0, 0, 1, {0: 5})]
    mm = {(0, 1): [(5, -1, 0.0), (5, 4, 0.0)], (0, 2): []}
    out = ao.find_2d3d_matches(3, {0}, cloud, mm)
    assert out == {1: [(0, 4)], 2: []}             # every not-done view gets an entry, even an empty one (SfM.cpp:524)


def test_find_only_the_upper_triangle_is_consulted():
    cloud = [P(0, 0, 1, {2: 5})]
    mm = {(2, 1): [(5, 9, 0.0)]}                   # stored the wrong way round: the reference indexes [1][2] only
    assert ao.find_2d3d_matches(3, {2}, cloud, mm)[1] == []


def test_cv_norm_is_float_difference_then_double_norm():
    a = np.array([1.0000001, 2.0, 3.0], np.float32)
    b = np.array([1.0, 2.0, 3.0], np.float32)
    d = np.float32(a[0] - b[0])
    assert ao.cv_norm_diff(a, b) == float(np.sqrt(np.float64(d) * np.float64(d)))


def test_merge_new_point_far_from_everything_is_appended():
    cloud = [P(0, 0, 1, {0: 1, 1: 1})]
    new = [P(5, 5, 5, {1: 7, 2: 7})]
    n, m, mmx = ao.merge_new_point_cloud(cloud, new, {})
    assert (n, m, mmx) == (1, 0, []) and len(cloud) == 2 and cloud[1][1] == {1: 7, 2: 7}


def test_merge_close_point_without_feature_match_is_dropped():
    cloud = [P(0, 0, 1, {0: 1, 1: 1})]
    new = [P(0, 0, 1.001, {1: 7, 2: 7})]
    n, m, _ = ao.merge_new_point_cloud(cloud, new, {(0, 1): [(1, 7, 30.0)]})     # distance 30 >= 20: not a feature match
    assert (n, m) == (0, 0) and len(cloud) == 1 and cloud[0][1] == {0: 1, 1: 1}


def test_merge_adds_views_and_stops_at_first_matching_existing_point():
    cloud = [P(0, 0, 1, {0: 1}), P(0, 0, 1.002, {0: 2})]
    new = [P(0, 0, 1.001, {1: 7, 2: 9})]
    mm = {(0, 1): [(2, 7, 1.0), (1, 7, 1.0)], (0, 2): [(1, 9, 5.0)]}
    n, m, pushed = ao.merge_new_point_cloud(cloud, new, mm)
    assert (n, m) == (0, 1)
    assert cloud[0][1] == {0: 1, 1: 7, 2: 9}          # both new views confirmed against existing view 0
    assert cloud[1][1] == {0: 2}                      # the second close point is never looked at (SfM.cpp:590-593)
    assert pushed == [(0, 1, 1), (0, 2, 0)]


def test_merge_visits_a_view_inserted_behind_the_iterator():
    # existing point {0: 1}; new point {1: 7, 3: 9}.  While new view 1 is checked against existing view 0 it is INSERTED
    # (key 1 > current key 0), so the same inner loop goes on to existing view 1 and compares new view 1 with itself
    # (pair (1,1), no matches).  Then new view 3 is checked against existing views 0 AND the freshly inserted 1.
    cloud = [P(0, 0, 1, {0: 1})]
    new = [P(0, 0, 1, {1: 7, 3: 9})]
    mm = {(0, 1): [(1, 7, 1.0)], (1, 3): [(7, 9, 1.0)]}      # view 3 only matches through the inserted view 1
    n, m, pushed = ao.merge_new_point_cloud(cloud, new, mm)
    assert (n, m) == (0, 1) and cloud[0][1] == {0: 1, 1: 7, 3: 9}
    assert pushed == [(0, 1, 0), (1, 3, 0)]


def test_merge_later_new_point_sees_points_appended_earlier_in_the_same_call():
    cloud = []
    new = [P(1, 1, 1, {0: 1, 1: 1}), P(1, 1, 1.001, {1: 1, 2: 4})]
    mm = {(1, 2): [(1, 4, 3.0)]}
    n, m, _ = ao.merge_new_point_cloud(cloud, new, mm)
    assert (n, m) == (1, 1) and len(cloud) == 1 and cloud[0][1] == {0: 1, 1: 1, 2: 4}


def test_merge_overwrites_an_existing_view_entry():
    cloud = [P(0, 0, 1, {0: 1, 2: 3})]
    new = [P(0, 0, 1, {2: 8})]
    mm = {(0, 2): [(1, 8, 1.0)]}
    ao.merge_new_point_cloud(cloud, new, mm)
    assert cloud[0][1] == {0: 1, 2: 8}                # originatingViews[newKv.first] = newKv.second (SfM.cpp:582)


def test_radius_candidates_include_earlier_new_points_and_are_ascending():
    ex = np.array([[0, 0, 0], [1, 0, 0], [0, 0, 0.005]], np.float32)
    nw = np.array([[0, 0, 0.001], [0, 0, 0.002], [9, 9, 9]], np.float32)
    ptr, idx = ao.radius_candidates(ex, nw)
    assert ptr.tolist() == [0, 2, 5, 5]
    assert idx.tolist() == [0, 2, 0, 2, 3]


def test_radius_threshold_is_strict_and_nan_never_matches():
    h = np.float32(0.01)
    ex = np.array([[0, 0, 0], [np.nan, 0, 0]], np.float32)
    nw = np.array([[h, 0, 0], [np.nextafter(h, np.float32(0)), 0, 0]], np.float32)
    ptr, idx = ao.radius_candidates(ex, nw)
    # new 0 is exactly at distance h from existing 0: not < h.  new 1 is one ulp closer: matches existing 0, and new 0
    assert ptr.tolist() == [0, 0, 2] and idx.tolist() == [0, 2]
