"""Wall time of the drop-in call itself -- sfmtoylib::SfMBundleAdjustmentUtils::adjustBundle() on the reference's
containers (std::map per point etc.) -- at a BASELINE config size (default cfg3).  SFMBA_SHIM_TIMING=1 prints it."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sfm_toy_library_amd as sfm
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
prob = sfm.make_problem(name)
os.environ["SFMBA_SHIM_TIMING"] = "1"
os.environ.setdefault("SFMBA_MAX_SECONDS", "0")
os.environ.setdefault("SFMBA_PRECISION", "f32j")
lib = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sfm-toy-library_amd", "host", "libsfmba_shim.so"))
c = np.array(sfm.synthetic.PRINCIPAL_POINT, dtype=np.float32)
n_views = prob.n_cam + 1
poses = np.zeros((n_views, 3, 4), dtype=np.float32)
poses[:prob.n_cam, :, :3] = sfm.synthetic.rotvec_to_matrix(prob.cam6[:, :3]); poses[:prob.n_cam, :, 3] = prob.cam6[:, 3:]
K = np.array([[prob.focal, 0, c[0]], [0, prob.focal, c[1]], [0, 0, 1]], dtype=np.float32)
# feature lists per view (camera-major order of the observations), feature index of every observation
order = np.argsort(prob.obs_cam, kind="stable")
feat_ptr = np.zeros(n_views + 1, dtype=np.int64); feat_ptr[1:prob.n_cam + 1] = np.cumsum(np.bincount(prob.obs_cam, minlength=prob.n_cam)); feat_ptr[-1] = feat_ptr[-2]
feat_xy = (prob.obs_xy[order].astype(np.float32) + c).astype(np.float32)
feat_idx = np.empty(prob.n_obs, dtype=np.int32); feat_idx[order] = (np.arange(prob.n_obs) - feat_ptr[prob.obs_cam[order]]).astype(np.int32)
view_ptr = np.zeros(prob.n_pt + 1, dtype=np.int64); view_ptr[1:] = np.cumsum(np.bincount(prob.obs_pt, minlength=prob.n_pt))
view_idx = prob.obs_cam.astype(np.int32)
fp, ip, lp = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
for r in range(3):
    P, KK, X = poses.copy(), K.copy(), prob.pt3.astype(np.float32).copy()
    t0 = time.perf_counter()
    lib.sfmba_shim_adjust_bundle(C.c_int(n_views), P.ctypes.data_as(fp), KK.ctypes.data_as(fp), C.c_int(prob.n_pt), X.ctypes.data_as(fp),
                                 view_ptr.ctypes.data_as(lp), view_idx.ctypes.data_as(ip), feat_idx.ctypes.data_as(ip),
                                 feat_ptr.ctypes.data_as(lp), np.ascontiguousarray(feat_xy).ctypes.data_as(fp))
    print("harness total %.1f ms (container build + adjustBundle + copy back); focal %.3f -> %.3f" % (1e3 * (time.perf_counter() - t0), K[0, 0], KK[0, 0]), flush=True)
