"""Problem dump format: what crosses the adjustBundle() boundary as flat arrays.

Little-endian binary, all arrays C-contiguous:
  magic   8 bytes  b"SFMBA001"
  int32   n_cam, n_pt ; int64 n_obs ; float64 focal
  float64 cam6[n_cam*6] ; float64 pt3[n_pt*3]
  int32   obs_cam[n_obs] ; int32 obs_pt[n_obs] ; float64 obs_xy[n_obs*2]
The C++ shim writes the same bytes when SFMBA_DUMP=<path> is set (host/SfMBundleAdjustmentUtils.cpp),
so a BA input captured on a machine that has OpenCV can be replayed here.

BAL text interchange ("Bundle Adjustment in the Large", grail.cs.washington.edu/projects/bal):
  <n_cam> <n_pt> <n_obs>
  n_obs lines      <camera index> <point index> <x> <y>
  9 n_cam lines    per camera: angle-axis (3), translation (3), focal f, radial distortion k1, k2
  3 n_pt lines     per point: X, Y, Z
BAL's camera looks down the NEGATIVE z axis:  P = R X + t,  p = -P / P.z,  p' = f r(p) p,  r = 1 + k1 |p|^2 + k2 |p|^4,
observations relative to the image centre.  The reference's model (BA.cpp:58-97) is  f (P.x / P.z, P.y / P.z) - obs  with ONE
focal shared by all cameras and no distortion.  Mapping (load_bal / save_bal):
  * pose: identical (same angle-axis / translation convention);
  * observations: negated (f P/P.z - (-u) = -(BAL residual) when r = 1: same cost);
  * focal: BAL has one per camera, the reference one for the whole problem -> the problem's focal is the mean (or the median,
    or a given value); the per-camera values are kept in meta["bal_focal"] and written back by save_bal if still present;
  * k1, k2: no counterpart -- DROPPED on import (kept in meta["bal_k"], written back verbatim by save_bal, zero otherwise).
    A BAL problem with non-zero distortion or differing focals therefore starts from a higher cost here: documented loss.
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


def load_bal(path, focal="mean"):
    """Reads a BAL text problem into a BAProblem of the reference's parameterisation (see the module docstring for the mapping
    and its losses).  focal: "mean" | "median" | a number."""
    with open(path, "r") as f:
        tok = f.read().split()
    n_cam, n_pt, n_obs = int(tok[0]), int(tok[1]), int(tok[2])
    need = 3 + 4 * n_obs + 9 * n_cam + 3 * n_pt
    if len(tok) < need:
        raise ValueError("truncated BAL file %s: %d tokens, expected %d" % (path, len(tok), need))
    obs = np.array(tok[3:3 + 4 * n_obs], dtype=np.float64).reshape(n_obs, 4)
    o = 3 + 4 * n_obs
    cams = np.array(tok[o:o + 9 * n_cam], dtype=np.float64).reshape(n_cam, 9)
    o += 9 * n_cam
    pts = np.array(tok[o:o + 3 * n_pt], dtype=np.float64).reshape(n_pt, 3)
    obs_cam, obs_pt = obs[:, 0].astype(np.int32), obs[:, 1].astype(np.int32)
    if n_obs and (obs_cam.min() < 0 or obs_cam.max() >= n_cam or obs_pt.min() < 0 or obs_pt.max() >= n_pt):
        raise ValueError("BAL observation index out of range in %s" % path)
    f_cam = cams[:, 6].copy()
    f0 = float(np.mean(f_cam)) if focal == "mean" else float(np.median(f_cam)) if focal == "median" else float(focal)
    meta = dict(path=str(path), format="bal", bal_focal=f_cam, bal_k=cams[:, 7:9].copy(),
                bal_loss=dict(distortion_dropped=bool(np.any(cams[:, 7:9] != 0.0)), focal_spread=float(f_cam.max() - f_cam.min()) if n_cam else 0.0))
    return BAProblem(np.ascontiguousarray(cams[:, :6]), np.ascontiguousarray(pts), f0, obs_cam, obs_pt,
                     np.ascontiguousarray(-obs[:, 2:4]), None, None, float("nan"), meta)


def save_bal(path, prob, cam6=None, pt3=None, focal=None):
    """Writes a BAProblem (optionally with solved parameters) as BAL text.  Every camera gets the shared focal unless the
    problem came from load_bal and its focal is unchanged (then the original per-camera focals and k1, k2 are written back)."""
    cam6 = np.asarray(prob.cam6 if cam6 is None else cam6, dtype=np.float64)
    pt3 = np.asarray(prob.pt3 if pt3 is None else pt3, dtype=np.float64)
    f = float(prob.focal if focal is None else focal)
    meta = prob.meta or {}
    f_cam = np.full(prob.n_cam, f)
    k = np.zeros((prob.n_cam, 2))
    if "bal_k" in meta and len(meta["bal_k"]) == prob.n_cam:
        k = np.asarray(meta["bal_k"], dtype=np.float64)
        if focal is None and "bal_focal" in meta:
            f_cam = np.asarray(meta["bal_focal"], dtype=np.float64)
    with open(path, "w") as out:
        out.write("%d %d %d\n" % (prob.n_cam, prob.n_pt, prob.n_obs))
        xy = -np.asarray(prob.obs_xy, dtype=np.float64)
        for c, i, (x, y) in zip(prob.obs_cam, prob.obs_pt, xy):
            out.write("%d %d %.16e %.16e\n" % (c, i, x, y))
        for j in range(prob.n_cam):
            for v in list(cam6[j]) + [f_cam[j], k[j, 0], k[j, 1]]:
                out.write("%.16e\n" % v)
        for i in range(prob.n_pt):
            for v in pt3[i]:
                out.write("%.16e\n" % v)
