"""The drop-in boundary end to end (-m gpu): the C++ shim with the REFERENCE signature
sfmtoylib::SfMBundleAdjustmentUtils::adjustBundle(PointCloud&, vector<Matx34f>&, Intrinsics&, const vector<Features>&)
(sfm-toy-library_amd/host/) driven through a flat-array harness, against the oracle's restatement of
the reference function (oracle.adjust_bundle, BA.cpp:99-222): float angle-axis marshalling, principal-point
subtraction, empty poses, write-back only on CONVERGENCE."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "sfm-toy-library_amd", "host", "libsfmba_shim.so")


def _containers(prob, sfm, n_extra_views=1):
    c = np.array(sfm.synthetic.PRINCIPAL_POINT, dtype=np.float32)
    n_views = prob.n_cam + n_extra_views
    poses = np.zeros((n_views, 3, 4), dtype=np.float32)
    R = sfm.synthetic.rotvec_to_matrix(prob.cam6[:, :3])
    poses[:prob.n_cam, :, :3] = R
    poses[:prob.n_cam, :, 3] = prob.cam6[:, 3:]
    K = np.array([[prob.focal, 0, c[0]], [0, prob.focal, c[1]], [0, 0, 1]], dtype=np.float32)
    feats = [[] for _ in range(n_views)]
    views = [dict() for _ in range(prob.n_pt)]
    for k in range(prob.n_obs):
        v, i = int(prob.obs_cam[k]), int(prob.obs_pt[k])
        views[i][v] = len(feats[v])
        feats[v].append(prob.obs_xy[k].astype(np.float32) + c)
    feats = [np.array(f, dtype=np.float32).reshape(-1, 2) for f in feats]
    return poses, K, prob.pt3.astype(np.float32), views, feats


def _call_shim(poses, K, points, views, feats):
    lib = C.CDLL(SHIM)
    poses = np.ascontiguousarray(poses, dtype=np.float32).copy()
    K = np.ascontiguousarray(K, dtype=np.float32).copy()
    points = np.ascontiguousarray(points, dtype=np.float32).copy()
    view_ptr = np.zeros(len(views) + 1, dtype=np.int64)
    vi, fi = [], []
    for i, m in enumerate(views):
        for v in sorted(m):
            vi.append(v)
            fi.append(m[v])
        view_ptr[i + 1] = len(vi)
    vi, fi = np.array(vi, dtype=np.int32), np.array(fi, dtype=np.int32)
    feat_ptr = np.zeros(len(feats) + 1, dtype=np.int64)
    for v, f in enumerate(feats):
        feat_ptr[v + 1] = feat_ptr[v] + len(f)
    feat_xy = np.ascontiguousarray(np.concatenate(feats), dtype=np.float32)
    fp, ip, lp = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    lib.sfmba_shim_adjust_bundle(C.c_int(poses.shape[0]), poses.ctypes.data_as(fp), K.ctypes.data_as(fp), C.c_int(points.shape[0]),
                                 points.ctypes.data_as(fp), view_ptr.ctypes.data_as(lp), vi.ctypes.data_as(ip), fi.ctypes.data_as(ip),
                                 feat_ptr.ctypes.data_as(lp), feat_xy.ctypes.data_as(fp))
    return poses, K, points


@pytest.mark.parametrize("name", ["tiny", "crazyhorse_like"])
def test_shim_matches_reference_restatement(sfm, oracle, name, monkeypatch):
    monkeypatch.setenv("SFMBA_MAX_SECONDS", "0")
    prob = sfm.make_problem(name)
    poses, K, pts, views, feats = _containers(prob, sfm)
    p_o, K_o, pts_o, summ = oracle.adjust_bundle(poses, K, pts, views, feats, sfm.SfmbaOptions.defaults(max_seconds=0.0))
    assert summ["termination_name"] == "CONVERGENCE"
    p_g, K_g, pts_g = _call_shim(poses, K, pts, views, feats)
    assert np.array_equal(p_g[-1], np.zeros((3, 4), np.float32))            # empty pose untouched
    assert K_g[0, 0] == K_g[1, 1] and K_g[0, 2] == K[0, 2] and K_g[1, 2] == K[1, 2]
    # float containers: agreement to float round-off of the written-back values
    assert np.allclose(K_g, K_o, rtol=2e-7, atol=0)
    assert np.allclose(p_g, p_o, rtol=0, atol=2e-6)
    assert np.allclose(pts_g, pts_o, rtol=0, atol=2e-6)
    assert not np.array_equal(pts_g, pts)


@pytest.mark.parametrize("linear", ["default", "cholesky", "auto", "pcg"])
def test_shim_with_sixty_views_all_linear_solvers(sfm, oracle, linear, monkeypatch):
    """60 views (reduced dimension 361): the shim's default is the reference's exact DENSE_SCHUR-equivalent Cholesky;
    SFMBA_LINEAR=auto|pcg opts into the block-Jacobi PCG branch a many-view caller would pick.  Both against the oracle's
    restatement of adjustBundle() (BA.cpp:99-222) to float round-off of the written-back containers."""
    monkeypatch.setenv("SFMBA_MAX_SECONDS", "0")
    if linear == "default":
        monkeypatch.delenv("SFMBA_LINEAR", raising=False)
    else:
        monkeypatch.setenv("SFMBA_LINEAR", linear)
    prob = sfm.make_problem("cfg2", n_cam=60, n_pt=2000, views=6, seed=4343)
    poses, K, pts, views, feats = _containers(prob, sfm, n_extra_views=2)
    p_o, K_o, pts_o, summ = oracle.adjust_bundle(poses, K, pts, views, feats, sfm.SfmbaOptions.defaults(max_seconds=0.0))
    assert summ["termination_name"] == "CONVERGENCE"
    p_g, K_g, pts_g = _call_shim(poses, K, pts, views, feats)
    assert np.array_equal(p_g[-1], np.zeros((3, 4), np.float32)) and np.array_equal(p_g[-2], np.zeros((3, 4), np.float32))
    assert not np.array_equal(pts_g, pts)
    assert np.allclose(K_g, K_o, rtol=2e-7, atol=0)
    assert np.allclose(p_g, p_o, rtol=0, atol=2e-6)
    assert np.allclose(pts_g, pts_o, rtol=0, atol=2e-6)


def test_shim_leaves_everything_untouched_without_convergence(sfm, monkeypatch, tmp_path):
    """BA.cpp:182-185: a point on the camera plane makes the first evaluation fail -> FAILURE -> no write-back."""
    monkeypatch.setenv("SFMBA_MAX_SECONDS", "0")
    dump = tmp_path / "ba_input.sfmba"
    monkeypatch.setenv("SFMBA_DUMP", str(dump))
    prob = sfm.make_problem("tiny")
    k0 = int(np.nonzero(prob.obs_cam == 0)[0][0])
    prob.pt3[prob.obs_pt[k0]] = (0.1, 0.2, -5.0)
    poses, K, pts, views, feats = _containers(prob, sfm)
    p_g, K_g, pts_g = _call_shim(poses, K, pts, views, feats)
    assert np.array_equal(p_g, poses) and np.array_equal(K_g, K) and np.array_equal(pts_g, pts)
    # the SFMBA_DUMP hook wrote exactly what crossed the boundary
    back = sfm.load_problem(dump)
    assert back.n_cam == poses.shape[0] and back.n_pt == prob.n_pt and back.n_obs == prob.n_obs
    assert np.allclose(back.obs_xy, prob.obs_xy, atol=1e-4)


@pytest.mark.parametrize("overlap", ["1", "0"])
def test_shim_resident_cache_follows_a_growing_reconstruction(sfm, oracle, monkeypatch, capfd, overlap):
    """The reference re-runs adjustBundle() after every added view (SfM.cpp:464-466) on a cloud that only grows.  The shim
    keeps the previous problem resident and appends the difference; every call must still equal a fresh run of the oracle's
    restatement of adjustBundle() on that call's containers.  Then a call that does NOT contain the previous one (a point
    lost a view) must fall back to a rebuild and still be right."""
    monkeypatch.setenv("SFMBA_MAX_SECONDS", "0")
    monkeypatch.setenv("SFMBA_SHIM_TIMING", "1")
    monkeypatch.delenv("SFMBA_SHIM_CACHE", raising=False)
    # overlap = 1: the lists that kept their length are walked and compared on worker threads WHILE the GPU solves (the default);
    # 0: everything is marshalled and compared before the solve.  Same results, same paths.
    monkeypatch.setenv("SFMBA_SHIM_OVERLAP", overlap)
    prob = sfm.make_problem("cfg2", n_cam=9, n_pt=500, views=(2, 5), seed=77)
    poses, K, pts, views, feats = _containers(prob, sfm)

    def call(n_registered):
        # views >= n_registered are not registered yet: empty poses, their observations not in the cloud; points with fewer
        # than two registered views are not in the cloud at all (appended at the END when they appear, like the reference does)
        keep = [i for i, v in enumerate(views) if sum(1 for k in v if k < n_registered) >= 2]
        return keep

    order = []                                   # cloud order = order of first appearance
    paths = []
    for n_reg in (4, 5, 6, 7, 8, 9):
        for i in call(n_reg):
            if i not in order:
                order.append(i)
        vs = [{k: f for k, f in views[i].items() if k < n_reg} for i in order]
        ps = poses.copy(); ps[n_reg:] = 0
        cloud = pts[order]
        p_o, K_o, pts_o, summ = oracle.adjust_bundle(ps, K, cloud, vs, feats, sfm.SfmbaOptions.defaults(max_seconds=0.0))
        assert summ["termination_name"] == "CONVERGENCE"
        p_g, K_g, pts_g = _call_shim(ps, K, cloud, vs, feats)
        err = capfd.readouterr().err
        paths.append([l.split("path:")[1].strip() for l in err.splitlines() if "path:" in l][-1])
        assert np.allclose(K_g, K_o, rtol=2e-7, atol=0)
        assert np.allclose(p_g, p_o, rtol=0, atol=2e-6) and np.allclose(pts_g, pts_o, rtol=0, atol=2e-6)
        assert np.array_equal(p_g[n_reg:], np.zeros_like(p_g[n_reg:]))
    assert paths[1:] == ["append"] * 5, paths
    # same containers again: nothing to add
    _call_shim(ps, K, cloud, vs, feats)
    assert "path: resident" in capfd.readouterr().err
    # a point loses a view: not a superset of the cached list -> rebuild, result still the oracle's
    vs2 = [dict(v) for v in vs]
    victim = next(i for i, v in enumerate(vs2) if len(v) >= 3)
    vs2[victim].pop(max(vs2[victim]))
    p_o, K_o, pts_o, summ = oracle.adjust_bundle(ps, K, cloud, vs2, feats, sfm.SfmbaOptions.defaults(max_seconds=0.0))
    p_g, K_g, pts_g = _call_shim(ps, K, cloud, vs2, feats)
    assert "path: rebuild" in capfd.readouterr().err
    assert np.allclose(p_g, p_o, rtol=0, atol=2e-6) and np.allclose(pts_g, pts_o, rtol=0, atol=2e-6)
    # a changed observation coordinate is caught as well
    feats2 = [f.copy() for f in feats]
    v0 = min(vs2[0]); feats2[v0][vs2[0][v0]] += 0.25
    p_o, K_o, pts_o, summ = oracle.adjust_bundle(ps, K, cloud, vs2, feats2, sfm.SfmbaOptions.defaults(max_seconds=0.0))
    p_g, K_g, pts_g = _call_shim(ps, K, cloud, vs2, feats2)
    assert "path: rebuild" in capfd.readouterr().err
    assert np.allclose(pts_g, pts_o, rtol=0, atol=2e-6)
    # ... and so is a point that traded one view for another (same list length, different view): call it twice so that the second
    # call starts from a resident problem built on feats2 / vs2
    _call_shim(ps, K, cloud, vs2, feats2)
    capfd.readouterr()
    vs3 = [dict(v) for v in vs2]
    victim = next(i for i, v in enumerate(vs3) if len(v) >= 3 and len(v) < 9)
    gone = max(vs3[victim]); new_view = next(k for k in range(9) if k not in vs3[victim])
    vs3[victim].pop(gone); vs3[victim][new_view] = 0
    p_o, K_o, pts_o, summ = oracle.adjust_bundle(ps, K, cloud, vs3, feats2, sfm.SfmbaOptions.defaults(max_seconds=0.0))
    p_g, K_g, pts_g = _call_shim(ps, K, cloud, vs3, feats2)
    assert "path: rebuild" in capfd.readouterr().err
    if summ["termination_name"] == "CONVERGENCE":
        assert np.allclose(p_g, p_o, rtol=0, atol=2e-6) and np.allclose(pts_g, pts_o, rtol=0, atol=2e-6)
    else:
        assert np.array_equal(pts_g, cloud.astype(np.float32))
