"""The streaming CG on ONE triangle of the preconditioned reduced matrix (-m gpu; round 6, ABI v6: sfmba_options.pcg_symmetric; csrc/dense_solver.hip
"Symmetric streaming path": k_sy_vec + k_sy_prod + k_sy_coarse).  The matrix in question is the dense S of the reference's DENSE_SCHUR (BA.cpp:172) after
the block-Jacobi transform; what is held here: the symmetric form reaches the ORACLE's result in both precisions, with and without the coarse space and
under AUTO (1e-12, no Cholesky fallback), takes the iterations the both-triangles kernels of round 5 take (pcg_symmetric = -1), and a deterministic
handle -- whose sums may not arrive through atomics -- stays bitwise repeatable on the old kernels."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from sfm_toy_library_amd import capi as c
    assert c.device_count() >= 1
    return c


@pytest.fixture(scope="module")
def prob(sfm):
    return sfm.make_problem("cfg3", n_cam=230, n_pt=6000, seed=77)          # d = 1381 > 1280: the streaming CG


@pytest.fixture(scope="module")
def want(sfm, oracle, prob):
    return oracle.solve(prob, sfm.SfmbaOptions.defaults(max_seconds=0.0))


def _close(got, want, cost_rtol, cam_atol):
    assert got[3]["termination_name"] == want[3]["termination_name"] == "CONVERGENCE"
    assert got[3]["iterations"] == want[3]["iterations"]
    assert abs(got[3]["final_cost"] - want[3]["final_cost"]) <= cost_rtol * want[3]["final_cost"]
    assert np.abs(got[0] - want[0]).max() < cam_atol and abs(got[2] - want[2]) < 1e3 * cam_atol


@pytest.mark.parametrize("precision,f32m,cost_rtol,cam_atol", [(0, 0, 1e-9, 1e-7), (1, 0, 1e-6, 2e-5), (1, -1, 1e-6, 2e-5)])
def test_symmetric_streaming_cg_matches_the_oracle_and_the_both_triangles_kernels(capi, prob, want, precision, f32m, cost_rtol, cam_atol):
    base = dict(max_seconds=0.0, precision=precision, linear_solver=1, pcg_tolerance=1e-10, pcg_anchored=0, pcg_f32_matrix=f32m)
    sym = capi.solve(prob, capi.default_options(**base))                       # library default: the one-triangle kernels
    on = capi.solve(prob, capi.default_options(pcg_symmetric=1, **base))
    full = capi.solve(prob, capi.default_options(pcg_symmetric=-1, **base))    # round 5: both triangles written and read
    for r in (sym, on, full):
        _close(r, want, cost_rtol, cam_atol)
    # same CG up to rounding: the iteration counts of a solve differ by a few at most, the results by the CG tolerance
    assert abs(sym[3]["linear_iters"] - full[3]["linear_iters"]) <= 2 * sym[3]["iterations"]
    assert abs(sym[3]["final_cost"] - full[3]["final_cost"]) <= (1e-11 if precision == 0 else 1e-7) * full[3]["final_cost"]
    assert np.abs(sym[0] - full[0]).max() < (1e-8 if precision == 0 else 2e-5)


def test_symmetric_cg_without_the_coarse_space(capi, prob, want):
    base = dict(max_seconds=0.0, precision=0, linear_solver=1, pcg_tolerance=1e-10, pcg_anchored=0, pcg_coarse_space=-1)
    sym = capi.solve(prob, capi.default_options(**base))
    full = capi.solve(prob, capi.default_options(pcg_symmetric=-1, **base))
    _close(sym, want, 1e-9, 1e-6)
    assert abs(sym[3]["linear_iters"] - full[3]["linear_iters"]) <= 3 * sym[3]["iterations"]
    two_level = capi.solve(prob, capi.default_options(max_seconds=0.0, precision=0, linear_solver=1, pcg_tolerance=1e-10, pcg_anchored=0))
    assert sym[3]["linear_iters"] > 1.3 * two_level[3]["linear_iters"]        # (the coarse space of the symmetric form does its work)


def test_auto_reaches_the_dense_schur_result_on_the_one_triangle(capi, prob, want):
    """SFMBA_LINEAR_AUTO = the DENSE_SCHUR result through the CG at 1e-12: no Cholesky fallback, the oracle's trajectory."""
    for precision, rtol, atol in ((0, 1e-9, 1e-8), (1, 1e-6, 2e-5)):
        got = capi.solve(prob, capi.default_options(max_seconds=0.0, precision=precision))
        _close(got, want, rtol, atol)
        assert got[3]["cholesky_fallbacks"] == 0 and got[3]["linear_iters"] > 0


def test_resident_problem_switches_between_the_forms(capi, prob, want):
    """One handle, alternating solves with and without the symmetric form (the pair pass then writes one triangle / both): nothing of one solve may leak
    into the next (the zeroed accumulation buffers, the lower triangle left over from an earlier solve)."""
    with capi.Problem(prob, precision=1) as P:
        costs = []
        for sw in (0, -1, 0, -1, 1):
            P.reset()
            s, _ = P.solve(capi.default_options(max_seconds=0.0, precision=1, linear_solver=1, pcg_symmetric=sw))
            assert s["termination_name"] == "CONVERGENCE" and s["iterations"] == want[3]["iterations"]
            costs.append(s["final_cost"])
        assert max(abs(c - want[3]["final_cost"]) for c in costs) <= 1e-6 * want[3]["final_cost"]


def test_deterministic_handle_keeps_the_both_triangles_kernels(capi, sfm, prob):
    """Atomics arrive in no fixed order: SFMBA_CREATE_DETERMINISTIC handles ignore pcg_symmetric and stay bitwise repeatable."""
    runs = []
    with capi.Problem(prob, precision=1, flags=sfm.CREATE_DETERMINISTIC) as P:
        for sw in (0, 1, -1, 0):
            P.reset()
            s, tr = P.solve(capi.default_options(max_seconds=0.0, precision=1, linear_solver=1, pcg_symmetric=sw))
            runs.append((P.get_params(), s["final_cost"], s["linear_iters"]))
    for (cam, pt, f), c, li in runs[1:]:
        assert np.array_equal(cam, runs[0][0][0]) and np.array_equal(pt, runs[0][0][1]) and f == runs[0][0][2] and c == runs[0][1] and li == runs[0][2]


def test_dimension_not_a_multiple_of_the_tile_and_tiny_last_strip(capi, sfm, oracle):
    """d = 6 * 214 + 1 = 1285: the last strip of 32 rows holds 5, the last column chunk 5 columns; 300 cameras: d = 1801."""
    for n_cam, seed in ((214, 11), (300, 12)):
        p = sfm.make_problem("cfg3", n_cam=n_cam, n_pt=4000, seed=seed)
        w = oracle.solve(p, sfm.SfmbaOptions.defaults(max_seconds=0.0))
        got = capi.solve(p, capi.default_options(max_seconds=0.0, precision=0, linear_solver=1, pcg_tolerance=1e-11, pcg_anchored=0))
        _close(got, w, 1e-9, 1e-7)
