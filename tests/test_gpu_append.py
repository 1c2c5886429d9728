"""SURVEY 8(f) row 1 (-m gpu): the incremental caller re-runs BA after every added view (SfM.cpp:464-466).  A resident
problem GROWS in place through sfmba_problem_append -- new cameras, new points, new views of existing points -- and at every
step must give what a fresh solve of the whole problem gives: compared with the ORACLE's solve of the step's problem
(same termination, iterations, cost, parameters) and with a freshly created HIP problem."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from sfm_toy_library_amd import capi as c
    assert c.device_count() >= 1
    return c


def incremental_steps(prob, first_cams, step=1):
    """Observation sets of a reconstruction that registers the cameras in index order: at a step with cameras < c the
    observations of every point seen by at least two of them.  Each step adds whole new points AND new views of old points."""
    steps, have = [], np.zeros(prob.n_obs, bool)
    for c in list(range(first_cams, prob.n_cam + 1, step)):
        vis = prob.obs_cam < c
        cnt = np.bincount(prob.obs_pt[vis], minlength=prob.n_pt)
        now = vis & (cnt[prob.obs_pt] >= 2)
        assert not np.any(have & ~now)                     # the set only grows
        new = np.flatnonzero(now & ~have)
        steps.append((c, new))
        have = now
    return steps


def step_problem(sfm, prob, order):
    return sfm.BAProblem(prob.cam6, prob.pt3, prob.focal, prob.obs_cam[order], prob.obs_pt[order], prob.obs_xy[order])


@pytest.mark.parametrize("precision,linear", [(0, 0), (1, 1)])
def test_incremental_sequence_matches_fresh_oracle_solves(capi, sfm, oracle, precision, linear):
    prob = sfm.make_problem("cfg2", n_cam=14, n_pt=900, views=(2, 6), seed=321)
    steps = incremental_steps(prob, first_cams=3)
    assert len(steps) >= 10
    opt = capi.default_options(max_seconds=0.0, precision=precision, linear_solver=linear)
    order = steps[0][1]
    P = capi.Problem(step_problem(sfm, prob, order), precision=precision)
    try:
        for si, (c, new) in enumerate(steps):
            if si > 0:
                assert len(new) > 0
                P.append(prob.cam6, prob.pt3, prob.focal, prob.obs_cam[new], prob.obs_pt[new], prob.obs_xy[new])
                order = np.concatenate([order, new])
            sub = step_problem(sfm, prob, order)
            assert P.n_obs == sub.n_obs and P.reduced_dim == 6 * len(np.unique(sub.obs_cam)) + 1
            want = oracle.solve(sub, sfm.SfmbaOptions.defaults(max_seconds=0.0))
            s, tr = P.solve(opt)
            cam, pt, f = P.get_params()
            assert s["termination_name"] == want[3]["termination_name"] == "CONVERGENCE"
            assert s["iterations"] == want[3]["iterations"], (si, s["iterations"], want[3]["iterations"])
            tol = 1e-9 if precision == 0 else 1e-6
            assert abs(s["final_cost"] - want[3]["final_cost"]) <= tol * want[3]["final_cost"]
            assert np.isclose(s["initial_cost"], want[3]["initial_cost"], rtol=1e-9)
            # F32J: the rounding of the fp32 Jacobian blocks moves a weakly held camera of these small sub-problems by 1e-5 .. 3e-5 at equal cost
            # (the cost to 1e-6 above; tests/f32j_accuracy.py prints the same figures for other shapes)
            atol = 1e-7 if precision == 0 else 5e-5
            assert np.abs(cam - want[0]).max() <= atol and np.abs(pt - want[1]).max() <= atol
            # residual vector in the CALLER's observation order (old observations first, then each append's)
            res, cost = P.eval_residuals()
            res_o, cost_o = oracle.eval_residuals(sub, cam, pt, f)
            assert np.allclose(res, res_o, rtol=1e-9, atol=1e-7)
            # ... and a freshly created problem on the same list
            fresh = capi.solve(sub, opt)
            assert fresh[3]["iterations"] == s["iterations"]
            assert abs(fresh[3]["final_cost"] - s["final_cost"]) <= 1e-9 * s["final_cost"]
            assert np.allclose(fresh[0], cam, atol=1e-7 if precision == 0 else 5e-6)
    finally:
        P.close()


def test_append_to_an_empty_problem_and_argument_checks(capi, sfm):
    prob = sfm.make_problem("tiny")
    empty = sfm.BAProblem(prob.cam6, prob.pt3, prob.focal, prob.obs_cam[:0], prob.obs_pt[:0], prob.obs_xy[:0])
    with capi.Problem(empty) as P:
        assert P.reduced_dim == 0
        P.append(prob.cam6, prob.pt3, prob.focal, prob.obs_cam, prob.obs_pt, prob.obs_xy)
        s, _ = P.solve(capi.default_options(max_seconds=0.0))
        ref = capi.solve(prob, capi.default_options(max_seconds=0.0))
        assert s["iterations"] == ref[3]["iterations"] and abs(s["final_cost"] - ref[3]["final_cost"]) <= 1e-12 * ref[3]["final_cost"]
        with pytest.raises(capi.SfmbaError):                  # arrays may only grow
            P.append(prob.cam6[:-1], prob.pt3, prob.focal, prob.obs_cam[:0], prob.obs_pt[:0], prob.obs_xy[:0])
        with pytest.raises(capi.SfmbaError):                  # index beyond the arrays
            P.append(prob.cam6, prob.pt3, prob.focal, np.array([prob.n_cam], np.int32), np.array([0], np.int32), np.zeros((1, 2)))
        # a failed call leaves the problem usable
        s2, _ = P.solve(capi.default_options(max_seconds=0.0))
        assert s2["termination_name"] == "CONVERGENCE"


def test_append_at_cfg3_scale_is_cheaper_than_a_rebuild(capi, sfm):
    """One more view on top of 199 (cfg 3): ~5000 new observations -- new views of existing points -- merged on the device."""
    import time
    prob = sfm.make_problem("cfg3")
    old = np.flatnonzero(prob.obs_cam < prob.n_cam - 1)
    new = np.flatnonzero(prob.obs_cam == prob.n_cam - 1)
    base = step_problem(sfm, prob, old)
    opt = capi.default_options(max_seconds=0.0, precision=1, linear_solver=1)
    with capi.Problem(base, precision=1) as P:
        P.solve(opt)
        t0 = time.perf_counter()
        P.append(prob.cam6, prob.pt3, prob.focal, prob.obs_cam[new], prob.obs_pt[new], prob.obs_xy[new])
        t_append = time.perf_counter() - t0
        s, _ = P.solve(opt)
    full = step_problem(sfm, prob, np.concatenate([old, new]))
    t0 = time.perf_counter()
    with capi.Problem(full, precision=1) as Q:
        t_create = time.perf_counter() - t0
        s2, _ = Q.solve(opt)
    assert s["iterations"] == s2["iterations"] and abs(s["final_cost"] - s2["final_cost"]) <= 1e-9 * s2["final_cost"]
    print("append %.2f ms vs create %.2f ms" % (1e3 * t_append, 1e3 * t_create))
    assert t_append < t_create
