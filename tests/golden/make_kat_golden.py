"""Regenerates tests/golden/reprojection_kat.json from the REFERENCE's own unit-test fixture.

Source of truth: /root/reference/SfMToyLib/SfMUnitTests.cpp
  :53-56   INTRINSICS f=700, c=(320,240)
  :59-71   cannedPoints3d (12 points)
  :80-95   mock camera: Euler (5,5,5) deg -> R (ceres::EulerAnglesToRotationMatrix, R=Rz(yaw)Ry(roll)Rx(pitch)),
           t=(-10,0,30); image points by cv::projectPoints (Rodrigues(R), t, K, no distortion)
  :153-189 ceres_reprojection_test: AngleAxisRotatePoint + t, divide, f*x + c ; tolerance 0.1 px
The expected pixels below are cv::projectPoints' pinhole formula u = f*(RX+t)_x/(RX+t)_z + c evaluated
in fp64 numpy (OpenCV itself is not installable here); the reference test accepts 0.1 px.
This script does not import the oracle.
"""
import json
import os
import numpy as np

POINTS = [(4, 12, 50), (12, 11, 55), (22, 1, 45), (13, 3, 60), (11, 16, 61), (21, 12, 65),
          (24, 11, 67), (29, 6, 41), (27, 4, 44), (22, 7, 58), (20, 9, 51), (15, 10, 40)]
F, C = 700.0, (320.0, 240.0)
EULER_DEG = (5.0, 5.0, 5.0)     # pitch, roll, yaw
T = (-10.0, 0.0, 30.0)


def euler_to_R(e):
    pitch, roll, yaw = np.deg2rad(e)
    c1, s1, c2, s2, c3, s3 = np.cos(yaw), np.sin(yaw), np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch)
    return np.array([[c1 * c2, -s1 * c3 + c1 * s2 * s3, s1 * s3 + c1 * s2 * c3],
                     [s1 * c2, c1 * c3 + s1 * s2 * s3, -c1 * s3 + s1 * s2 * c3],
                     [-s2, c2 * s3, c2 * c3]])


def main():
    R = euler_to_R(EULER_DEG)
    X = np.array(POINTS, dtype=np.float64)
    p = X @ R.T + np.array(T)
    uv = F * p[:, :2] / p[:, 2:3] + np.array(C)
    # angle-axis of R (fp64, via the log map) for reference
    theta = np.arccos((np.trace(R) - 1.0) / 2.0)
    w = theta / (2.0 * np.sin(theta)) * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    out = dict(source="SfMToyLib/SfMUnitTests.cpp:53-95,153-189", focal=F, principal_point=C,
               euler_deg=EULER_DEG, translation=T, tolerance_px=0.1,
               R=R.tolist(), angle_axis=w.tolist(), points3d=POINTS, pixels=uv.tolist())
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reprojection_kat.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(path)
    print(np.round(uv, 6))


if __name__ == "__main__":
    main()
