"""Wall time of the drop-in adjustBundle() in the reference's call pattern: BA re-run after every added view (SfM.cpp:464-466).
cfg 3 with 199 registered views, then the 200th view is added (its ~5000 observations are new views of existing points):
the second call goes through the shim's resident-problem cache (sfmba_problem_append).  SFMBA_SHIM_TIMING=1 prints the split."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sfm_toy_library_amd as sfm
args = [a for a in sys.argv[1:] if not a.startswith("--")]
prob = sfm.make_problem(args[0] if args else "cfg3")
if "--warmup" in sys.argv:
    # sfmba_device_warmup (include/sfmba.h): the process-wide first-call costs paid before the first adjustBundle()
    # (through the C ABI directly, as a C++ host would: capi.lib() would import torch first)
    hip = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sfm-toy-library_amd", "csrc", "libsfmba_hip.so"), mode=C.RTLD_GLOBAL)
    t0 = time.perf_counter()
    rc = hip.sfmba_device_warmup(C.c_int(0), C.c_int64(prob.n_obs))
    print("sfmba_device_warmup(0, %d): rc %d, %.1f ms" % (prob.n_obs, rc, 1e3 * (time.perf_counter() - t0)), flush=True)
os.environ["SFMBA_SHIM_TIMING"] = "1"
os.environ.setdefault("SFMBA_MAX_SECONDS", "0")
os.environ.setdefault("SFMBA_PRECISION", "f32j")
os.environ.setdefault("SFMBA_LINEAR", "pcg")
lib = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sfm-toy-library_amd", "host", "libsfmba_shim.so"))
c = np.array(sfm.synthetic.PRINCIPAL_POINT, dtype=np.float32)
n_views = prob.n_cam
poses = np.zeros((n_views, 3, 4), dtype=np.float32)
poses[:, :, :3] = sfm.synthetic.rotvec_to_matrix(prob.cam6[:, :3]); poses[:, :, 3] = prob.cam6[:, 3:]
K = np.array([[prob.focal, 0, c[0]], [0, prob.focal, c[1]], [0, 0, 1]], dtype=np.float32)
order = np.argsort(prob.obs_cam, kind="stable")
feat_ptr = np.zeros(n_views + 1, dtype=np.int64); feat_ptr[1:] = np.cumsum(np.bincount(prob.obs_cam, minlength=prob.n_cam))
feat_xy = np.ascontiguousarray((prob.obs_xy[order].astype(np.float32) + c).astype(np.float32))
feat_idx_all = np.empty(prob.n_obs, dtype=np.int32); feat_idx_all[order] = (np.arange(prob.n_obs) - feat_ptr[prob.obs_cam[order]]).astype(np.int32)
fp, ip, lp = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int64)


def call(n_reg, label):
    sel = prob.obs_cam < n_reg
    view_ptr = np.zeros(prob.n_pt + 1, dtype=np.int64); view_ptr[1:] = np.cumsum(np.bincount(prob.obs_pt[sel], minlength=prob.n_pt))
    view_idx = np.ascontiguousarray(prob.obs_cam[sel].astype(np.int32)); feat_idx = np.ascontiguousarray(feat_idx_all[sel])
    P = poses.copy(); P[n_reg:] = 0
    KK, X = K.copy(), prob.pt3.astype(np.float32).copy()
    t0 = time.perf_counter()
    lib.sfmba_shim_adjust_bundle(C.c_int(n_views), P.ctypes.data_as(fp), KK.ctypes.data_as(fp), C.c_int(prob.n_pt), X.ctypes.data_as(fp),
                                 view_ptr.ctypes.data_as(lp), view_idx.ctypes.data_as(ip), feat_idx.ctypes.data_as(ip),
                                 feat_ptr.ctypes.data_as(lp), feat_xy.ctypes.data_as(fp))
    print("%s: harness total %.1f ms; focal %.3f -> %.3f" % (label, 1e3 * (time.perf_counter() - t0), K[0, 0], KK[0, 0]), flush=True)


for rep in range(2):
    call(n_views - 1, "%d views (rep %d)" % (n_views - 1, rep))
    call(n_views, "%d views: one view added" % n_views)
