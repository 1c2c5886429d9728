"""GPU parity tests (-m gpu): every call goes through the C ABI (include/sfmba.h) into the HIP
kernels and is compared with the CPU oracle on the same seeded inputs and with the committed
golden fixtures.  Tolerances: fp64 mode -> rounding level; F32J mode (fp32 Jacobian blocks,
fp64 residuals/accumulation) -> the 1e-4 px RMS bar of BASELINE.json."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def capi():
    from sfm_toy_library_amd import capi as c
    assert c.device_count() >= 1, "no HIP device: the GPU tests must run on the MI355X box"
    return c


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(GOLD, "solver_golden.json")) as f:
        return json.load(f)


def _opts(sfm, **kw):
    kw.setdefault("max_seconds", 0.0)
    return sfm.SfmbaOptions.defaults(**kw)


def rms(cost, n_obs):
    return np.sqrt(2.0 * cost / n_obs)


# ---------------------------------------------------------------------------------------------
# model: residuals and analytic Jacobian blocks vs the oracle's Jet autodiff
# ---------------------------------------------------------------------------------------------
def test_reference_kat_fixture_on_gpu(capi, sfm):
    """SfMUnitTests.cpp:153-189 fixture through the HIP residual kernel: r == 0."""
    with open(os.path.join(GOLD, "reprojection_kat.json")) as f:
        kat = json.load(f)
    cam = np.array([kat["angle_axis"] + kat["translation"]])
    pts = np.array(kat["points3d"], dtype=np.float64)
    c = np.array(kat["principal_point"])
    obs = np.array(kat["pixels"]) - c
    n = len(pts)
    prob = sfm.BAProblem(cam, pts, kat["focal"], np.zeros(n, np.int32), np.arange(n, dtype=np.int32), obs)
    with capi.Problem(prob, precision=0) as P:
        res, cost = P.eval_residuals()
    assert np.max(np.abs(res)) < 1e-9
    assert cost < 1e-18


@pytest.mark.parametrize("name", ["tiny", "small", "cfg2"])
def test_residuals_match_oracle(capi, sfm, oracle, name):
    prob = sfm.make_problem(name)
    res_o, cost_o = oracle.eval_residuals(prob)
    for precision in (0, 1):
        with capi.Problem(prob, precision=precision) as P:
            res, cost = P.eval_residuals()
        assert np.allclose(res, res_o, rtol=1e-11, atol=1e-9)      # residuals are fp64 in both modes
        assert np.isclose(cost, cost_o, rtol=1e-12)


@pytest.mark.parametrize("name", ["tiny", "small", "cfg2"])
def test_jacobian_blocks_match_oracle_jets(capi, sfm, oracle, name):
    prob = sfm.make_problem(name)
    _, jc_o, jp_o, jf_o = oracle.eval_jacobian(prob)
    with capi.Problem(prob, precision=0) as P:
        jc, jp, jf = P.eval_jacobian()
    scale = max(np.abs(jc_o).max(), 1.0)
    assert np.allclose(jc, jc_o, rtol=1e-9, atol=1e-9 * scale)
    assert np.allclose(jp, jp_o, rtol=1e-9, atol=1e-9 * scale)
    assert np.allclose(jf, jf_o, rtol=1e-12, atol=1e-12)
    with capi.Problem(prob, precision=1) as P:
        jc, jp, jf = P.eval_jacobian()
    assert np.allclose(jc, jc_o, rtol=2e-4, atol=2e-5 * scale)      # fp32 blocks
    assert np.allclose(jp, jp_o, rtol=2e-4, atol=2e-5 * scale)


def test_jacobian_special_rotations(capi, sfm, oracle):
    """theta = 0 (first-order branch, as the reference's first camera), tiny theta, theta near pi."""
    cams = np.array([[0, 0, 0, 0, 0, 5.0], [1e-9, -2e-9, 1e-9, 0.1, 0, 5.0], [3.0, 0.4, 0.1, -1.0, 0.3, 5.0],
                     [2.0, 1.5, -1.2, 0.5, 0.2, 6.0]])
    pts = np.array([[0.3, 0.7, -0.4], [-0.9, 0.1, 0.8], [0.2, -0.2, 0.9]])
    oc, op = np.meshgrid(np.arange(4), np.arange(3), indexing="ij")
    oc, op = oc.ravel().astype(np.int32), op.ravel().astype(np.int32)
    prob = sfm.BAProblem(cams, pts, 1234.5, oc, op, np.zeros((12, 2)))
    res_o, jc_o, jp_o, jf_o = oracle.eval_jacobian(prob)
    with capi.Problem(prob, precision=0) as P:
        res, _ = P.eval_residuals()
        jc, jp, jf = P.eval_jacobian()
    assert np.allclose(res, res_o, rtol=1e-12, atol=1e-9)
    s = np.abs(jc_o).max()
    assert np.allclose(jc, jc_o, rtol=1e-8, atol=1e-9 * s)
    assert np.allclose(jp, jp_o, rtol=1e-8, atol=1e-9 * s)


# ---------------------------------------------------------------------------------------------
# reduced camera system (Schur complement incl. the shared-focal border)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,precision,tol", [("tiny", 0, 1e-10), ("small", 0, 1e-10), ("cfg2", 0, 1e-9), ("small", 1, 2e-4)])
def test_reduced_system_matches_oracle(capi, sfm, oracle, name, precision, tol):
    prob = sfm.make_problem(name)
    radius = 1e4
    S_o, rhs_o, scale_o, info = oracle.build_reduced(prob, radius)
    assert info == 0
    with capi.Problem(prob, precision=precision) as P:
        S, rhs, scale = P.build_reduced(radius)
    assert np.allclose(scale, scale_o, rtol=max(tol, 1e-12))
    assert np.allclose(S, S.T)
    assert np.abs(S - S_o).max() <= tol * np.abs(S_o).max()
    assert np.abs(rhs - rhs_o).max() <= tol * np.abs(rhs_o).max()


def test_reduced_system_without_jacobi_scaling(capi, sfm, oracle):
    prob = sfm.make_problem("tiny")
    opt = _opts(sfm, jacobi_scaling=0)
    S_o, rhs_o, scale_o, _ = oracle.build_reduced(prob, 50.0, opt)
    with capi.Problem(prob) as P:
        S, rhs, scale = P.build_reduced(50.0, opt)
    assert np.all(scale == 1.0)
    assert np.abs(S - S_o).max() <= 1e-10 * np.abs(S_o).max()
    assert np.abs(rhs - rhs_o).max() <= 1e-10 * np.abs(rhs_o).max()


# ---------------------------------------------------------------------------------------------
# dense reduced-system solver in isolation
# ---------------------------------------------------------------------------------------------
# 1215 / 1216: the augmented row fills the last tile / needs a tile of its own; 2559 / 2561: last size of the one-launch-per-block-column
# factorisation (40 block columns) / first size of the two-kernel form
@pytest.mark.parametrize("n", [1, 7, 63, 64, 121, 200, 1201, 1215, 1216, 2559, 2561])
def test_dense_cholesky(capi, n):
    rng = np.random.default_rng(n)
    M = rng.normal(size=(n, n))
    A = M @ M.T + n * np.eye(n)
    b = rng.normal(size=n)
    x, info, _ = capi.dense_spd_solve(A, b, method=0)
    assert info == 0
    ref = np.linalg.solve(A, b)
    assert np.allclose(x, ref, rtol=1e-9, atol=1e-11 * np.abs(ref).max())


@pytest.mark.parametrize("cond", [1e4, 1e8, 1e11])
def test_dense_cholesky_ill_conditioned(capi, cond):
    """The fused factorisation solves the panel with the explicit inverse of the 64 x 64 diagonal factor (a GEMM on the matrix
    cores) instead of a triangular sweep: the backward error must stay at the level of a plain Cholesky (numpy's, here) also when
    the reduced system is as badly conditioned as a gauge-free BA makes it (1e8 and beyond)."""
    n = 640
    rng = np.random.default_rng(int(np.log10(cond)))
    Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    A = (Q * np.logspace(0, -np.log10(cond), n)) @ Q.T
    A = 0.5 * (A + A.T)
    b = A @ rng.normal(size=n)
    x, info, _ = capi.dense_spd_solve(A, b, method=0)
    assert info == 0
    ref = np.linalg.solve(A, b)
    res = np.linalg.norm(A @ x - b) / np.linalg.norm(b)
    res_ref = np.linalg.norm(A @ ref - b) / np.linalg.norm(b)
    assert res <= max(50 * res_ref, 1e-13), (res, res_ref)


@pytest.mark.parametrize("n", [2561, 2700])
def test_dense_cholesky_beyond_the_fused_form(capi, n):
    """More than 40 block columns of 64 (d > 2560): the two-kernel factorisation and the step-by-step back substitution take over -- chosen
    by size alone (ABI v4 had SFMBA_CHOL_* environment switches to force them at any size; nothing below sfmba_problem_create* reads the
    environment any more)."""
    rng = np.random.default_rng(5)
    M = rng.normal(size=(n, n))
    A = M @ M.T + n * np.eye(n)
    b = rng.normal(size=n)
    x, info, _ = capi.dense_spd_solve(A, b, method=0)
    ref = np.linalg.solve(A, b)
    assert info == 0 and np.allclose(x, ref, rtol=1e-9, atol=1e-11 * np.abs(ref).max()), np.abs(x - ref).max()


@pytest.mark.parametrize("n", [40, 100, 700])        # one launch (d < 64) / several block columns
def test_dense_cholesky_flags_indefinite(capi, n):
    A = np.eye(n) * 4
    A[37, 37] = -1.0
    x, info, _ = capi.dense_spd_solve(A, np.ones(n), method=0)
    assert info == 38          # leading minor of order 38 is not positive definite


@pytest.mark.parametrize("n", [7, 121, 1201, 1501, 3002])     # > 1280: the generic (streaming) CG iteration kernel
def test_dense_pcg(capi, n):
    rng = np.random.default_rng(100 + n)
    M = rng.normal(size=(n, n)) / np.sqrt(n)
    A = M @ M.T + np.eye(n)
    b = rng.normal(size=n)
    x, info, iters = capi.dense_spd_solve(A, b, method=1, tol=1e-12)
    assert info == 0 and iters > 0
    assert np.linalg.norm(A @ x - b) <= 1e-10 * np.linalg.norm(b)


# ---------------------------------------------------------------------------------------------
# full LM solves vs the oracle and the committed golden results
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["tiny", "small", "crazyhorse_like", "cfg2"])
def test_solve_matches_oracle_fp64(capi, sfm, oracle, golden, name):
    prob = sfm.make_problem(name)
    cam_o, pt_o, f_o, s_o, tr_o = oracle.solve(prob, _opts(sfm))
    cam, pt, f, s, tr = capi.solve(prob, capi.default_options(max_seconds=0.0))
    assert s["termination_name"] == s_o["termination_name"] == golden[name]["termination"]
    assert s["iterations"] == s_o["iterations"] == golden[name]["iterations"]
    assert np.isclose(s["initial_cost"], s_o["initial_cost"], rtol=1e-12)
    # BASELINE.json config 2: parity to 1e-6 (relative) in cost
    assert abs(s["final_cost"] - s_o["final_cost"]) <= 1e-6 * s_o["final_cost"]
    assert abs(s["final_cost"] - golden[name]["final_cost"]) <= 1e-6 * golden[name]["final_cost"]
    assert abs(rms(s["final_cost"], prob.n_obs) - golden[name]["rms_px"]) < 1e-4
    # same LM trajectory: per-iteration cost and radius
    assert len(tr) == len(tr_o)
    for a, b in zip(tr, tr_o):
        assert a["step_is_successful"] == b["step_is_successful"]
        assert np.isclose(a["cost"], b["cost"], rtol=1e-7)
        assert np.isclose(a["trust_region_radius"], b["trust_region_radius"], rtol=1e-4)
    # the returned parameters reproduce the reported cost (checked by the oracle's residual)
    _, c = oracle.eval_residuals(prob, cam, pt, f)
    assert np.isclose(c, s["final_cost"], rtol=1e-10)
    assert np.isclose(f, f_o, rtol=1e-7)


@pytest.mark.parametrize("fname", ["tiny.sfmba", "small.sfmba", "small_far.sfmba"])
def test_solve_committed_dumps(capi, sfm, golden, fname):
    prob = sfm.load_problem(os.path.join(GOLD, fname))
    g = golden[fname.split(".")[0]]
    cam, pt, f, s, tr = capi.solve(prob, capi.default_options(max_seconds=0.0))
    assert s["termination_name"] == g["termination"] and s["iterations"] == g["iterations"]
    assert abs(s["final_cost"] - g["final_cost"]) <= 1e-6 * g["final_cost"]
    assert np.allclose([r["cost"] for r in tr], g["trace_cost"], rtol=1e-7)
    assert np.isclose(f, g["focal"], rtol=1e-7)


@pytest.mark.parametrize("name", ["small", "crazyhorse_like", "cfg2"])
def test_solve_f32_jacobians_within_rms_bar(capi, sfm, golden, name):
    """BASELINE.json config 3 precision (fp32 Jacobians + fp64 accumulate): final RMS within 1e-4 px."""
    prob = sfm.make_problem(name)
    cam, pt, f, s, tr = capi.solve(prob, capi.default_options(max_seconds=0.0, precision=1))
    assert s["termination_name"] == "CONVERGENCE"
    assert abs(rms(s["final_cost"], prob.n_obs) - golden[name]["rms_px"]) < 1e-4
    assert abs(s["final_cost"] - golden[name]["final_cost"]) <= 1e-5 * golden[name]["final_cost"]


@pytest.mark.parametrize("name", ["small", "cfg2"])
def test_solve_pcg_matches_cholesky(capi, sfm, golden, name):
    prob = sfm.make_problem(name)
    cam, pt, f, s, tr = capi.solve(prob, capi.default_options(max_seconds=0.0, linear_solver=1, pcg_tolerance=1e-12, pcg_anchored=0))
    assert s["termination_name"] == "CONVERGENCE" and s["linear_iters"] > 0
    assert s["iterations"] == golden[name]["iterations"]
    assert abs(s["final_cost"] - golden[name]["final_cost"]) <= 1e-6 * golden[name]["final_cost"]


def test_resident_problem_reset_and_resolve(capi, sfm, golden):
    prob = sfm.make_problem("small")
    with capi.Problem(prob) as P:
        s1, _ = P.solve(capi.default_options(max_seconds=0.0))
        c1, p1, f1 = P.get_params()
        P.reset()
        c0, p0, f0 = P.get_params()
        assert np.array_equal(c0, prob.cam6) and np.array_equal(p0, prob.pt3) and f0 == prob.focal
        s2, _ = P.solve(capi.default_options(max_seconds=0.0))
        c2, p2, f2 = P.get_params()
        assert s1["iterations"] == s2["iterations"]
        assert np.isclose(s1["final_cost"], s2["final_cost"], rtol=1e-12)
        assert np.allclose(c1, c2, rtol=0, atol=1e-9) and np.allclose(p1, p2, rtol=0, atol=1e-9)
        # solving again from the converged point terminates immediately-ish and does not increase the cost
        s3, _ = P.solve(capi.default_options(max_seconds=0.0))
        assert s3["final_cost"] <= s2["final_cost"] * (1 + 1e-12)


# ---------------------------------------------------------------------------------------------
# edge cases the reference's behaviour defines (BA.cpp:118-122,160,174-176,182-185)
# ---------------------------------------------------------------------------------------------
def test_iteration_limit_gives_no_convergence(capi, sfm, oracle):
    prob = sfm.make_problem("tiny")
    cam, pt, f, s, tr = capi.solve(prob, capi.default_options(max_seconds=0.0, max_iters=1))
    cam_o, pt_o, f_o, s_o, _ = oracle.solve(prob, _opts(sfm, max_iters=1))
    assert s["termination_name"] == "NO_CONVERGENCE" and s["iterations"] == 1
    assert np.isclose(s["final_cost"], s_o["final_cost"], rtol=1e-9)
    assert np.allclose(cam, cam_o, atol=1e-9) and np.allclose(pt, pt_o, atol=1e-9)


def test_unreferenced_blocks_untouched(capi, sfm):
    prob = sfm.make_problem("tiny")
    cam6 = np.vstack([prob.cam6, np.zeros((2, 6))])
    pt3 = np.vstack([prob.pt3, [[9.0, 9.0, 9.0]]])
    big = sfm.BAProblem(cam6, pt3, prob.focal, prob.obs_cam, prob.obs_pt, prob.obs_xy)
    cam_a, pt_a, f_a, s_a, _ = capi.solve(prob, capi.default_options(max_seconds=0.0))
    cam_b, pt_b, f_b, s_b, _ = capi.solve(big, capi.default_options(max_seconds=0.0))
    assert np.array_equal(cam_b[-2:], np.zeros((2, 6))) and np.array_equal(pt_b[-1], [9.0, 9.0, 9.0])
    assert np.allclose(cam_a, cam_b[:-2], atol=1e-10) and np.isclose(f_a, f_b, rtol=1e-12)


def test_shuffled_observation_order_is_equivalent(capi, sfm):
    prob = sfm.make_problem("small")
    perm = np.random.default_rng(7).permutation(prob.n_obs)
    shuf = sfm.BAProblem(prob.cam6, prob.pt3, prob.focal, prob.obs_cam[perm], prob.obs_pt[perm], prob.obs_xy[perm])
    with capi.Problem(shuf) as P:
        res, cost = P.eval_residuals()
    with capi.Problem(prob) as P:
        res0, cost0 = P.eval_residuals()
    assert np.array_equal(res, res0[perm])
    a = capi.solve(prob, capi.default_options(max_seconds=0.0))
    b = capi.solve(shuf, capi.default_options(max_seconds=0.0))
    assert a[3]["iterations"] == b[3]["iterations"] and np.isclose(a[3]["final_cost"], b[3]["final_cost"], rtol=1e-10)


def test_point_on_camera_plane_is_failure(capi, sfm):
    prob = sfm.make_problem("tiny")
    k0 = int(np.nonzero(prob.obs_cam == 0)[0][0])
    prob.pt3[prob.obs_pt[k0]] = (0.1, 0.2, -5.0)          # camera 0 is R=I, t=(0,0,5): p_z = 0 exactly
    cam, pt, f, s, tr = capi.solve(prob, capi.default_options(max_seconds=0.0))
    assert s["termination_name"] == "FAILURE"
    assert np.array_equal(cam, prob.cam6) and np.array_equal(pt, prob.pt3) and f == prob.focal


def test_empty_problem(capi, sfm):
    prob = sfm.make_problem("tiny")
    empty = sfm.BAProblem(prob.cam6, prob.pt3, prob.focal, prob.obs_cam[:0], prob.obs_pt[:0], prob.obs_xy[:0])
    cam, pt, f, s, tr = capi.solve(empty)
    assert s["termination_name"] == "CONVERGENCE" and s["iterations"] == 0 and np.array_equal(cam, prob.cam6)


def test_duplicate_camera_point_pairs(capi, sfm, oracle):
    """The C ABI allows the same camera to observe a point twice (std::map in the reference cannot)."""
    prob = sfm.make_problem("tiny")
    extra = np.arange(0, prob.n_obs, 5)
    dup = sfm.BAProblem(prob.cam6, prob.pt3, prob.focal, np.concatenate([prob.obs_cam, prob.obs_cam[extra]]),
                        np.concatenate([prob.obs_pt, prob.obs_pt[extra]]),
                        np.concatenate([prob.obs_xy, prob.obs_xy[extra] + 0.25]))
    S_o, rhs_o, _, _ = oracle.build_reduced(dup, 100.0)
    with capi.Problem(dup) as P:
        S, rhs, _ = P.build_reduced(100.0)
    assert np.abs(S - S_o).max() <= 1e-10 * np.abs(S_o).max()
    s_o = oracle.solve(dup, _opts(sfm))[3]
    s = capi.solve(dup, capi.default_options(max_seconds=0.0))[3]
    assert s["iterations"] == s_o["iterations"] and abs(s["final_cost"] - s_o["final_cost"]) <= 1e-6 * s_o["final_cost"]


def test_cross_lane_exchanges_on_the_hardware():
    """csrc/sfmba_device.h builds every wave reduction on DPP permutes and v_permlane16/32_swap; tools/micro/permtest.hip checks each helper
    lane by lane against the plain definition on the device (the binary is built by __graft_entry__.build())."""
    import os, subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "micro", "permtest")
    if not os.path.exists(exe):
        pytest.skip("tools/micro/permtest not built")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
