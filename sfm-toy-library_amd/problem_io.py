"""Problem dump format: what crosses the adjustBundle() boundary as flat arrays.

Little-endian binary, all arrays C-contiguous:
  magic   8 bytes  b"SFMBA001"
  int32   n_cam, n_pt ; int64 n_obs ; float64 focal
  float64 cam6[n_cam*6] ; float64 pt3[n_pt*3]
  int32   obs_cam[n_obs] ; int32 obs_pt[n_obs] ; float64 obs_xy[n_obs*2]
The C++ shim writes the same bytes when SFMBA_DUMP=<path> is set (host/SfMBundleAdjustmentUtils.cpp),
so a BA input captured on a machine that has OpenCV can be replayed here.
"""
import struct
import numpy as np
from .synthetic import BAProblem

MAGIC = b"SFMBA001"


def save_problem(path, prob):
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<iiqd", prob.n_cam, prob.n_pt, prob.n_obs, float(prob.focal)))
        f.write(np.ascontiguousarray(prob.cam6, dtype="<f8").tobytes())
        f.write(np.ascontiguousarray(prob.pt3, dtype="<f8").tobytes())
        f.write(np.ascontiguousarray(prob.obs_cam, dtype="<i4").tobytes())
        f.write(np.ascontiguousarray(prob.obs_pt, dtype="<i4").tobytes())
        f.write(np.ascontiguousarray(prob.obs_xy, dtype="<f8").tobytes())


def load_problem(path):
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError("not an SFMBA001 problem dump: %s" % path)
        n_cam, n_pt, n_obs, focal = struct.unpack("<iiqd", f.read(24))
        cam6 = np.frombuffer(f.read(48 * n_cam), dtype="<f8").reshape(n_cam, 6).copy()
        pt3 = np.frombuffer(f.read(24 * n_pt), dtype="<f8").reshape(n_pt, 3).copy()
        obs_cam = np.frombuffer(f.read(4 * n_obs), dtype="<i4").copy()
        obs_pt = np.frombuffer(f.read(4 * n_obs), dtype="<i4").copy()
        obs_xy = np.frombuffer(f.read(16 * n_obs), dtype="<f8").reshape(n_obs, 2).copy()
    return BAProblem(cam6, pt3, focal, obs_cam, obs_pt, obs_xy, None, None, float("nan"), dict(path=str(path)))
