"""The triangulation oracle against the reference's own known-answer test (triangulate_from_2_views,
SfMToyLib/SfMUnitTests.cpp:221-251; fixture tests/golden/stereo_kat.json made by tests/golden/make_stereo_golden.py)."""
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stereo_kat.json")


def test_oracle_recovers_the_canned_points():
    from oracle import triangulate_oracle as tri
    g = json.load(open(GOLD))
    X, keep, el, er = tri.triangulate_views(g["K"], g["P_left"], g["P_right"], g["left"], g["right"])
    assert keep.all()
    assert np.linalg.norm(X.astype(np.float64) - np.array(g["points3d"], dtype=np.float64), axis=1).max() < g["tolerance"]
    assert el.max() < 1e-2 and er.max() < 1e-2


def test_oracle_drops_points_with_large_reprojection_error():
    from oracle import triangulate_oracle as tri
    g = json.load(open(GOLD))
    right = np.array(g["right"], dtype=np.float32)
    right[3] += (0.0, 60.0)           # a vertical disparity no 3D point can explain: > 10 px in at least one view
    X, keep, el, er = tri.triangulate_views(g["K"], g["P_left"], g["P_right"], g["left"], right)
    assert not keep[3] and keep.sum() == len(keep) - 1
