"""Camera counts beyond anything else in the suite (-m gpu; VERDICT r5 "weak" 10 / item 8c).  The reference hands ANY number of views to DENSE_SCHUR
(BA.cpp:142-172); here the reduced matrix of d = 6 n + 1 rows is indexed on the device, and d * ld crosses 2^31 at ~7 700 cameras -- every index product
of the pair pass, the CG kernels (symmetric streaming path: 134 k tiles) and the blocked factorisation (732 block columns) is exercised by the 7 800-camera
case.  No oracle at these sizes (its dense LLT would take hours): three independent paths of the product must agree -- F32J + CG on the fp32 triangle,
fp64 + CG, fp64 + factorisation."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from sfm_toy_library_amd import capi as c
    assert c.device_count() >= 1
    return c


def _run(capi, prob, precision, linear, max_iters):
    with capi.Problem(prob, precision=precision) as P:
        s, tr = P.solve(capi.default_options(max_seconds=0.0, precision=precision, linear_solver=linear, max_iters=max_iters))
        cam, pt, f = P.get_params()
    capi.release_cache()
    return cam, pt, f, s, tr


def _agree(a, b, cost_rtol, cam_atol):
    assert a[3]["iterations"] == b[3]["iterations"] and a[3]["termination_name"] == b[3]["termination_name"]
    assert abs(a[3]["final_cost"] - b[3]["final_cost"]) <= cost_rtol * b[3]["final_cost"]
    assert np.abs(a[0] - b[0]).max() < cam_atol and abs(a[2] - b[2]) < 1e-5 * abs(b[2])


def test_2500_cameras_three_paths_agree(capi, sfm):
    prob = sfm.make_problem("cfg3_banded", n_cam=2500, n_pt=15000, seed=4242)          # d = 15 001
    chol = _run(capi, prob, 0, 0, 30)
    assert chol[3]["termination_name"] == "CONVERGENCE" and chol[3]["final_cost"] < 1e-3 * chol[3]["initial_cost"]
    pcg = _run(capi, prob, 0, 1, 30)
    f32 = _run(capi, prob, 1, 1, 30)
    _agree(pcg, chol, 1e-9, 1e-5)
    _agree(f32, chol, 1e-6, 1e-4)
    rms = np.sqrt(2 * chol[3]["final_cost"] / prob.n_obs)
    assert 0.3 < rms < 1.0                                       # the 0.5 px noise of the generator, not a wrong minimum


def test_7800_cameras_index_products_beyond_2_to_31(capi, sfm):
    prob = sfm.make_problem("cfg3_banded", n_cam=7800, n_pt=30000, seed=4242)
    d = 6 * prob.n_cam + 1
    assert d * ((d + 1 + 63) // 64 * 64) > 2**31
    # three LM iterations each (a full solve of this path-shaped problem is ~30 s of CG at eight global coarse vectors; the segments stop at 8 192 unknowns)
    chol = _run(capi, prob, 0, 0, 3)
    pcg = _run(capi, prob, 0, 1, 3)
    f32 = _run(capi, prob, 1, 1, 3)
    assert chol[3]["iterations"] == 3 and chol[3]["final_cost"] < 1e-3 * chol[3]["initial_cost"]
    for r in (chol, pcg, f32):
        assert np.isfinite(r[0]).all() and np.isfinite(r[1]).all()
        assert [t["step_is_successful"] for t in r[4][1:]] == [1, 1, 1]
    _agree(pcg, chol, 1e-8, 1e-4)
    _agree(f32, chol, 1e-6, 5e-4)
