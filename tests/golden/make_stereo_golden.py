"""Regenerates tests/golden/stereo_kat.json from the REFERENCE's triangulation unit-test fixture.

Source of truth: /root/reference/SfMToyLib/SfMUnitTests.cpp
  :53-56    INTRINSICS f=700, c=(320,240)
  :59-71    cannedPoints3d (12 points)
  :105-150  generateStereoViews: left camera Euler (5,5,5) deg, t=(-10,0,30); right camera Euler (-5,0,5) deg, t=(10,0,28)
            (ceres::EulerAnglesToRotationMatrix<float>, R = Rz(yaw) Ry(roll) Rx(pitch)); image points by cv::projectPoints
  :221-251  triangulate_from_2_views: every triangulated point within 0.01 of its canned 3D point
Poses are rounded through fp32 like the reference's Matx34f; pixels are the pinhole projection evaluated in fp64 and rounded
to fp32 (Points2f).  OpenCV itself is not installable here.  This script does not import the oracle.
"""
import json
import os
import numpy as np

from make_kat_golden import POINTS, F, C, euler_to_R


def view(euler, t):
    R = euler_to_R(euler).astype(np.float32).astype(np.float64)
    t = np.array(t, dtype=np.float64)
    X = np.array(POINTS, dtype=np.float64)
    p = X @ R.T + t
    uv = (F * p[:, :2] / p[:, 2:3] + np.array(C)).astype(np.float32)
    P = np.concatenate([R, t[:, None]], axis=1).astype(np.float32)
    return P, uv


def main():
    Pl, uvl = view((5.0, 5.0, 5.0), (-10.0, 0.0, 30.0))
    Pr, uvr = view((-5.0, 0.0, 5.0), (10.0, 0.0, 28.0))
    out = dict(source="SfMToyLib/SfMUnitTests.cpp:105-150,221-251", K=[[F, 0, C[0]], [0, F, C[1]], [0, 0, 1]],
               P_left=Pl.tolist(), P_right=Pr.tolist(), left=uvl.tolist(), right=uvr.tolist(), points3d=POINTS,
               tolerance=0.01, max_reprojection_px=10.0)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stereo_kat.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(path)


if __name__ == "__main__":
    main()
