"""saveCloudAndCamerasToPLY (SfM.cpp:631-711): the C++ writer of host/SfMExport.cpp against the oracle's byte-exact
restatement, and the restatement against a hand-written expected file.  Pure host code (no GPU needed)."""
import ctypes as C
import os

import numpy as np

from oracle import ply_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "sfm-toy-library_amd", "host", "libsfmba_shim.so")


def test_oracle_against_a_hand_written_file():
    img = np.zeros((4, 6, 3), np.uint8)
    img[2, 3] = (10, 20, 30)                          # B, G, R at row 2, col 3
    img[0, 0] = (1, 2, 3)
    feats = [np.array([[2.5, 1.5], [0.4, 0.49]], np.float32)]          # (2.5, 1.5) -> col 2 (half to even), row 2; (0.4, 0.49) -> (0, 0)
    feats[0][0] = (3.4, 2.4)
    cloud = [(np.array([1.5, -2.25, 1e-7], np.float32), {0: 0}), (np.array([123456.789, 0.1, 3.0], np.float32), {0: 1})]
    text = ply_oracle.points_ply(cloud, feats, [img])
    assert text.splitlines()[:3] == ["ply                 ", "format ascii 1.0    ", "element vertex 2"]
    assert text.splitlines()[-2:] == ["1.5 -2.25 1e-07 30 20 10 ", "123457 0.1 3 3 2 1 "]
    assert text.endswith(" \n")
    pose = np.array([[[1, 0, 0, 0.5], [0, 1, 0, -1], [0, 0, 1, 2]]], np.float32)
    cams = ply_oracle.cameras_ply(pose).splitlines()
    assert cams[2] == "element vertex 4" and cams[6] == "element edge 3"
    assert cams[13:17] == ["0.5 -1 2", "0.7 -1 2", "0.5 -0.8 2", "0.5 -1 2.2"]
    assert cams[17:] == ["0 1 255 0 0", "0 2 0 255 0", "0 3 0 0 255"]


def test_cpp_writer_is_byte_identical(tmp_path, sfm):
    rng = np.random.default_rng(3)
    prob = sfm.make_problem("small")
    n_views, rows, cols = prob.n_cam, 48, 64
    images = rng.integers(0, 256, (n_views, rows, cols, 3), dtype=np.uint8)
    feats = [np.stack([rng.uniform(0, cols - 1, 40), rng.uniform(0, rows - 1, 40)], 1).astype(np.float32) for _ in range(n_views)]
    feats[0][0] = (10.5, 11.5)                        # round-half-even cases: 10.5 -> 10, 11.5 -> 12
    views = []
    for i in range(prob.n_pt):
        vs = np.unique(prob.obs_cam[prob.obs_pt == i])
        views.append({int(v): int(rng.integers(0, 40)) for v in vs})
    views[0] = {0: 0}
    pts = (prob.pt3 * np.array([1.0, 1e-5, 1e6])).astype(np.float32)           # exercises the %g exponent branches
    cloud = [(pts[i], views[i]) for i in range(prob.n_pt)]
    poses = np.zeros((n_views, 3, 4), np.float32)
    poses[:, :, :3] = sfm.synthetic.rotvec_to_matrix(prob.cam6[:, :3]); poses[:, :, 3] = prob.cam6[:, 3:]
    want_points, want_cams = ply_oracle.points_ply(cloud, feats, list(images)), ply_oracle.cameras_ply(poses)

    lib = C.CDLL(SHIM)
    vp = np.zeros(prob.n_pt + 1, np.int64); vi, fi = [], []
    for i, m in enumerate(views):
        for v in sorted(m):
            vi.append(v); fi.append(m[v])
        vp[i + 1] = len(vi)
    vi, fi = np.array(vi, np.int32), np.array(fi, np.int32)
    fptr = np.zeros(n_views + 1, np.int64); fptr[1:] = np.cumsum([len(f) for f in feats])
    fxy = np.ascontiguousarray(np.concatenate(feats), np.float32)
    ip, lp, fp = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_float)
    A = lambda a, tp: a.ctypes.data_as(tp)
    prefix = str(tmp_path / "recon")
    poses_c, pts_c, img_c = np.ascontiguousarray(poses), np.ascontiguousarray(pts), np.ascontiguousarray(images)
    rc = lib.sfmba_shim_save_ply(prefix.encode(), C.c_int(n_views), A(poses_c, fp), C.c_int(prob.n_pt), A(pts_c, fp), A(vp, lp), A(vi, ip), A(fi, ip),
                                 A(fptr, lp), A(fxy, fp), C.c_int(rows), C.c_int(cols), A(img_c, C.POINTER(C.c_ubyte)))
    assert rc == 0
    assert open(prefix + "_points.ply", "rb").read() == want_points.encode()
    assert open(prefix + "_cameras.ply", "rb").read() == want_cams.encode()
