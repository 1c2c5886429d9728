"""CPU restatement of SfMStereoUtilities::triangulateViews for ALIGNED matches -- test infrastructure only.

Follows /root/reference/SfMToyLib/SfMStereoUtilities.cpp:120-206 step by step; the OpenCV calls it makes are restated
from their published algorithms [OpenCV-upstream: calib3d undistortPoints / triangulatePoints (cvTriangulatePoints) /
convertPointsFromHomogeneous / projectPoints], with the float roundings where the reference's containers are float:

  :145-149  undistortPoints(points, K, no distortion)      x_n = (u - cx) / fx, y_n = (v - cy) / fy           -> float
  :151-152  triangulatePoints(Pleft, Pright, x_l, x_r)      DLT: A = [x P3 - P1; y P3 - P2] for both views (4x4, double),
                                                            SVD, last right singular vector                   -> float (4)
  :154-155  convertPointsFromHomogeneous                     X = (x, y, z) / w                                  -> float
  :157-169  Rodrigues(R) + projectPoints(X, rvec, t, K)      u = fx (R X + t)_x / (R X + t)_z + cx  (double)   -> float
  :183-190  keep the point iff BOTH reprojection errors are <= MIN_REPROJECTION_ERROR = 10 px (:42)

Pinned by the reference's own test triangulate_from_2_views (SfMUnitTests.cpp:221-251, tests/golden/stereo_kat.json): every
triangulated point within 0.01 of the canned 3D point.  Only tests/ may import this module.
"""
import numpy as np

MIN_REPROJECTION_ERROR = 10.0


def triangulate_views(K, P_left, P_right, left_xy, right_xy, max_err=MIN_REPROJECTION_ERROR):
    """Returns (points3d float32 [n,3], keep bool [n], err_left float64 [n], err_right float64 [n])."""
    K = np.asarray(K, dtype=np.float32).astype(np.float64)
    Pl = np.asarray(P_left, dtype=np.float32).astype(np.float64)
    Pr = np.asarray(P_right, dtype=np.float32).astype(np.float64)
    l = np.asarray(left_xy, dtype=np.float32).reshape(-1, 2)
    r = np.asarray(right_xy, dtype=np.float32).reshape(-1, 2)
    n = l.shape[0]
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]

    def normalise(p):
        q = np.stack([(p[:, 0].astype(np.float64) - cx) / fx, (p[:, 1].astype(np.float64) - cy) / fy], axis=1)
        return q.astype(np.float32).astype(np.float64)

    nl, nr = normalise(l), normalise(r)
    A = np.empty((n, 4, 4))
    A[:, 0] = nl[:, 0:1] * Pl[2] - Pl[0]
    A[:, 1] = nl[:, 1:2] * Pl[2] - Pl[1]
    A[:, 2] = nr[:, 0:1] * Pr[2] - Pr[0]
    A[:, 3] = nr[:, 1:2] * Pr[2] - Pr[1]
    if n:
        _, _, Vt = np.linalg.svd(A)
        Xh = Vt[:, 3, :].astype(np.float32)
    else:
        Xh = np.zeros((0, 4), dtype=np.float32)
    w = Xh[:, 3:4]
    scale = np.where(w != 0, np.float32(1.0) / np.where(w != 0, w, np.float32(1.0)), np.float32(1.0)).astype(np.float32)
    X = (Xh[:, :3] * scale).astype(np.float32)

    def project(P):
        p = X.astype(np.float64) @ P[:, :3].T + P[:, 3]
        with np.errstate(divide="ignore", invalid="ignore"):
            uv = np.stack([fx * p[:, 0] / p[:, 2] + cx, fy * p[:, 1] / p[:, 2] + cy], axis=1)
        return uv.astype(np.float32)

    el = np.linalg.norm((project(Pl) - l).astype(np.float64), axis=1)
    er = np.linalg.norm((project(Pr) - r).astype(np.float64), axis=1)
    keep = ~((el > max_err) | (er > max_err))         # NaN errors compare False in the reference's test too: kept
    return X, keep, el, er
