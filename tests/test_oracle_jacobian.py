"""The oracle's Jet autodiff (restating ceres::AutoDiffCostFunction<...,2,6,3,1>, BA.cpp:92) against
two independent derivatives: central finite differences and torch.autograd in fp64."""
import numpy as np
import pytest
import torch


def _residual_np(oracle, x, ox, oy):
    from ctypes import c_double
    r, _, _, _ = oracle.residual_jacobian(x[:6], x[6:9], x[9], ox, oy)
    return r


def _torch_residual(x, ox, oy):
    w, t, X, f = x[:3], x[3:6], x[6:9], x[9]
    theta2 = (w * w).sum()
    if theta2.item() > np.finfo(np.float64).eps:
        theta = torch.sqrt(theta2)
        k = w / theta
        p = X * torch.cos(theta) + torch.linalg.cross(k, X) * torch.sin(theta) + k * (k @ X) * (1 - torch.cos(theta))
    else:
        p = X + torch.linalg.cross(w, X)
    p = p + t
    return torch.stack([f * p[0] / p[2] - ox, f * p[1] / p[2] - oy])


CASES = [
    ("generic", [0.3, -0.2, 0.5, 0.1, -0.3, 4.0], [0.4, -0.6, 0.2], 2500.0),
    ("zero_rotation", [0.0, 0.0, 0.0, 0.0, 0.0, 5.0], [0.3, 0.7, -0.4], 2550.0),
    ("tiny_rotation", [1e-9, -2e-9, 1e-9, 0.0, 0.0, 5.0], [0.3, 0.7, -0.4], 2550.0),
    ("large_rotation", [2.0, 1.5, -1.2, 0.5, 0.2, 6.0], [-0.9, 0.1, 0.8], 700.0),
    ("near_pi", [3.0, 0.4, 0.1, -1.0, 0.3, 5.0], [0.2, -0.2, 0.9], 1000.0),
]


@pytest.mark.parametrize("name,cam,pt,focal", CASES)
def test_jets_vs_torch_autograd(oracle, name, cam, pt, focal):
    ox, oy = 12.5, -40.25
    r, jc, jp, jf = oracle.residual_jacobian(cam, pt, focal, ox, oy)
    x = torch.tensor(cam + pt + [focal], dtype=torch.float64, requires_grad=True)
    J = torch.autograd.functional.jacobian(lambda v: _torch_residual(v, ox, oy), x).numpy()
    rt = _torch_residual(x, ox, oy).detach().numpy()
    assert np.allclose(r, rt, rtol=1e-13, atol=1e-10)
    Jo = np.concatenate([jc, jp, jf.reshape(2, 1)], axis=1)
    assert np.allclose(Jo, J, rtol=1e-10, atol=1e-9), name


@pytest.mark.parametrize("name,cam,pt,focal", [c for c in CASES if c[0] not in ("tiny_rotation",)])
def test_jets_vs_finite_differences(oracle, name, cam, pt, focal):
    ox, oy = 3.0, 4.0
    x0 = np.array(cam + pt + [focal], dtype=np.float64)
    _, jc, jp, jf = oracle.residual_jacobian(cam, pt, focal, ox, oy)
    Jo = np.concatenate([jc, jp, jf.reshape(2, 1)], axis=1)
    if name == "zero_rotation":
        # the derivative Ceres' autodiff yields at theta=0 is that of the first-order branch X + w x X;
        # a central difference straddles the exact Rodrigues branch: same to O(h^2)
        pass
    J = np.zeros((2, 10))
    for k in range(10):
        h = 1e-6 * max(1.0, abs(x0[k]))
        xp, xm = x0.copy(), x0.copy()
        xp[k] += h
        xm[k] -= h
        J[:, k] = (_residual_np(oracle, xp, ox, oy) - _residual_np(oracle, xm, ox, oy)) / (2 * h)
    assert np.allclose(Jo, J, rtol=2e-6, atol=2e-5), name


def test_batched_matches_single(oracle, sfm):
    prob = sfm.make_problem("tiny")
    res, jc, jp, jf = oracle.eval_jacobian(prob)
    res2, cost = oracle.eval_residuals(prob)
    # Jet division is f.a * (1/g.a) (as in ceres/jet.h): values agree with the T=double path to an ulp or two
    assert np.allclose(res, res2, rtol=1e-13, atol=1e-11)
    assert np.isclose(cost, 0.5 * np.sum(res2 ** 2), rtol=1e-14)
    for k in (0, prob.n_obs // 2, prob.n_obs - 1):
        r, a, b, g = oracle.residual_jacobian(prob.cam6[prob.obs_cam[k]], prob.pt3[prob.obs_pt[k]], prob.focal,
                                              prob.obs_xy[k, 0], prob.obs_xy[k, 1])
        assert np.array_equal(r, res[k]) and np.array_equal(a, jc[k]) and np.array_equal(b, jp[k]) and np.array_equal(g, jf[k])
    # numpy projection used by the generator agrees with the functor
    uv, _ = sfm.synthetic.project(prob.cam6, prob.pt3, prob.focal, prob.obs_cam, prob.obs_pt)
    assert np.allclose(uv - prob.obs_xy, res, atol=1e-9)
