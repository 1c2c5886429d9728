"""HIP triangulation (sfmba_triangulate, csrc/triangulate.hip) vs the oracle restatement of
SfMStereoUtilities::triangulateViews and vs the reference's own known-answer test (-m gpu)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stereo_kat.json")


@pytest.fixture(scope="module")
def capi():
    import sfm_toy_library_amd  # noqa: F401
    from sfm_toy_library_amd import capi as c
    c.lib()
    return c


@pytest.fixture(scope="module")
def tri():
    from oracle import triangulate_oracle
    return triangulate_oracle


def test_reference_kat_on_gpu(capi):
    """triangulate_from_2_views (SfMUnitTests.cpp:221-251): every point within 0.01 of its canned 3D point."""
    g = json.load(open(GOLD))
    X, keep, err = capi.triangulate(g["K"], g["P_left"], g["P_right"], g["left"], g["right"])
    assert keep.all()
    assert np.linalg.norm(X.astype(np.float64) - np.array(g["points3d"], dtype=np.float64), axis=1).max() < g["tolerance"]
    assert err.max() < 1e-2


def _random_scene(n, seed, outlier_frac=0.1):
    rng = np.random.default_rng(seed)
    import sfm_toy_library_amd as sfm
    K = np.array([[2500.0, 0, 512.0], [0, 2500.0, 384.0], [0, 0, 1]], dtype=np.float32)
    Rl = np.eye(3)
    Rr = sfm.synthetic.rotvec_to_matrix(np.array([[0.02, -0.15, 0.01]]))[0]
    Pl = np.concatenate([Rl, np.zeros((3, 1))], axis=1).astype(np.float32)
    Pr = np.concatenate([Rr, np.array([[-1.0], [0.02], [0.1]])], axis=1).astype(np.float32)
    X = np.stack([rng.uniform(-1, 1, n), rng.uniform(-0.8, 0.8, n), rng.uniform(4, 8, n)], axis=1)

    def proj(P):
        p = X @ P[:, :3].astype(np.float64).T + P[:, 3].astype(np.float64)
        return (np.stack([K[0, 0] * p[:, 0] / p[:, 2] + K[0, 2], K[1, 1] * p[:, 1] / p[:, 2] + K[1, 2]], axis=1)
                + rng.normal(0, 0.5, (n, 2))).astype(np.float32)
    l, r = proj(Pl), proj(Pr)
    bad = rng.random(n) < outlier_frac
    r[bad] += rng.normal(0, 40.0, (int(bad.sum()), 2)).astype(np.float32)      # mismatches
    return K, Pl, Pr, l, r, X


@pytest.mark.parametrize("n,seed", [(1, 1), (63, 2), (5000, 3), (200000, 4)])
def test_matches_oracle(capi, tri, n, seed):
    K, Pl, Pr, l, r, _ = _random_scene(n, seed)
    X_o, keep_o, el_o, er_o = tri.triangulate_views(K, Pl, Pr, l, r)
    X, keep, err = capi.triangulate(K, Pl, Pr, l, r)
    # same points (float containers: a few ulp of the coordinates, which are O(1..10))
    assert np.allclose(X, X_o, rtol=2e-5, atol=2e-5)
    assert np.allclose(err[:, 0], el_o, rtol=1e-3, atol=2e-3) and np.allclose(err[:, 1], er_o, rtol=1e-3, atol=2e-3)
    # same keep decisions except where an error sits within float round-off of the 10 px threshold
    border = (np.abs(el_o - 10.0) < 5e-3) | (np.abs(er_o - 10.0) < 5e-3)
    assert np.array_equal(keep[~border], keep_o[~border])
    if n >= 5000:
        assert 0.02 * n < (~keep).sum() < 0.2 * n      # the planted mismatches are rejected, the inliers kept


def test_empty_and_bad_arguments(capi):
    K = np.eye(3, dtype=np.float32); P = np.zeros((3, 4), dtype=np.float32)
    X, keep, err = capi.triangulate(K, P, P, np.zeros((0, 2)), np.zeros((0, 2)))
    assert X.shape == (0, 3) and keep.shape == (0,)


def test_shim_with_the_reference_signature(tri):
    """sfmtoylib::SfMStereoUtilities::triangulateViews(Intrinsics, ImagePair, Matching, Features, Features, Matx34f, Matx34f,
    PointCloud&) through the flat-array harness: unaligned matches (queryIdx / trainIdx), back references, filter."""
    import ctypes as C
    shim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sfm-toy-library_amd", "host", "libsfmba_shim.so")
    lib = C.CDLL(shim)
    K, Pl, Pr, l, r, _ = _random_scene(3000, 9)
    rng = np.random.default_rng(5)
    perm_l, perm_r = rng.permutation(len(l)), rng.permutation(len(r))
    feats_l, feats_r = l[perm_l], r[perm_r]                      # features in arbitrary order ...
    q = np.argsort(perm_l).astype(np.int32); t = np.argsort(perm_r).astype(np.int32)   # ... match i = (q[i], t[i])
    sel = rng.permutation(len(l))[:2500]                          # not every feature is matched
    q, t = np.ascontiguousarray(q[sel]), np.ascontiguousarray(t[sel])
    X_o, keep_o, el_o, er_o = tri.triangulate_views(K, Pl, Pr, feats_l[q], feats_r[t])
    cap = len(q)
    X = np.zeros((cap, 3), np.float32); lr = np.zeros(cap, np.int32); rr = np.zeros(cap, np.int32)
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
    Kf, Plf, Prf = [np.ascontiguousarray(a, np.float32) for a in (K, Pl, Pr)]
    fl, fr = np.ascontiguousarray(feats_l, np.float32), np.ascontiguousarray(feats_r, np.float32)
    n = lib.sfmba_shim_triangulate_views(Kf.ctypes.data_as(fp), C.c_int(0), C.c_int(1), C.c_int(len(fl)), fl.ctypes.data_as(fp),
                                         C.c_int(len(fr)), fr.ctypes.data_as(fp), C.c_int(len(q)), q.ctypes.data_as(ip), t.ctypes.data_as(ip),
                                         Plf.ctypes.data_as(fp), Prf.ctypes.data_as(fp), C.c_int(cap), X.ctypes.data_as(fp),
                                         lr.ctypes.data_as(ip), rr.ctypes.data_as(ip))
    border = (np.abs(el_o - 10.0) < 5e-3) | (np.abs(er_o - 10.0) < 5e-3)
    assert not border.any()                                       # seed chosen so that no error sits on the threshold
    assert n == int(keep_o.sum())
    assert np.array_equal(lr[:n], q[keep_o]) and np.array_equal(rr[:n], t[keep_o])     # back references, match order
    assert np.allclose(X[:n], X_o[keep_o], rtol=2e-5, atol=2e-5)
