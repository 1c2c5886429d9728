"""ABI v4 / v5 (-m gpu): the behaviour switches that used to be environment variables only are fields of sfmba_options / create flags
(0 = library default, 1 = on, -1 = off; since ABI v5 NOTHING below sfmba_problem_create* reads the environment any more).  Every switch is driven THROUGH THE
C ABI here, with the environment clean, and must (a) act -- the observable the switch controls changes -- and (b) leave the result
where the reference algorithm puts it (the oracle, or the default configuration)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SWITCH_ENVS = ["SFMBA_PCG_COARSE", "SFMBA_PCG_PERSISTENT", "SFMBA_PCG_F32_MATRIX", "SFMBA_EARLY_LINEARISE", "SFMBA_SHARD_TWO_PHASE",
               "SFMBA_SHARD_F32_EXCHANGE", "SFMBA_DETERMINISTIC", "SFMBA_SHARD_DIST_CG", "SFMBA_PCG_SEGMENTS"]


@pytest.fixture(autouse=True)
def clean_env(monkeypatch):
    for e in SWITCH_ENVS:
        monkeypatch.delenv(e, raising=False)


@pytest.fixture(scope="module")
def capi():
    from sfm_toy_library_amd import capi as c
    assert c.device_count() >= 1
    return c


@pytest.fixture(scope="module")
def mid(sfm):
    """60 cameras (d = 361 > 256: the CG branch of AUTO), 8000 points."""
    return sfm.make_problem("cfg3", n_cam=60, n_pt=8000, seed=5)


@pytest.fixture(scope="module")
def mid_oracle(sfm, oracle, mid):
    return oracle.solve(mid, sfm.SfmbaOptions.defaults(max_seconds=0.0))


def same(got, want, atol, cost_rtol=1e-9):
    assert got[3]["termination_name"] == want[3]["termination_name"] == "CONVERGENCE"
    assert got[3]["iterations"] == want[3]["iterations"]
    assert abs(got[3]["final_cost"] - want[3]["final_cost"]) <= cost_rtol * want[3]["final_cost"]
    assert np.abs(got[0] - want[0]).max() <= atol and np.abs(got[1] - want[1]).max() <= atol


def test_default_is_auto_and_equals_dense_schur(capi, sfm, mid, mid_oracle):
    """The library default: the reference's DENSE_SCHUR result (BA.cpp:172) -- through the CG at 1e-12 above 256 unknowns."""
    o = capi.default_options(max_seconds=0.0)
    assert o.linear_solver == sfm.LINEAR_AUTO
    got = capi.solve(mid, o)
    same(got, mid_oracle, atol=1e-8)
    assert got[3]["linear_iters"] > 0 and got[3]["cholesky_fallbacks"] == 0          # it WAS the CG
    chol = capi.solve(mid, capi.default_options(max_seconds=0.0, linear_solver=sfm.LINEAR_CHOLESKY))
    assert chol[3]["linear_iters"] == 0
    assert np.abs(got[0] - chol[0]).max() < 1e-9 and np.abs(got[1] - chol[1]).max() < 1e-9
    # up to 256 unknowns AUTO factorises (the reference's own data sets)
    small = sfm.make_problem("cfg2")
    s = capi.solve(small, capi.default_options(max_seconds=0.0))[3]
    assert s["linear_iters"] == 0 and s["termination_name"] == "CONVERGENCE"


@pytest.mark.parametrize("precision", [0, 1])
def test_auto_falls_back_to_cholesky_when_the_cg_runs_out(capi, sfm, mid, mid_oracle, precision):
    """pcg_max_iters = 3 starves the CG in every LM iteration: each one must be re-solved by the Cholesky on the SAME linearisation
    (off-diagonal blocks re-formed unpreconditioned), and the result must be the exact path's."""
    o = capi.default_options(max_seconds=0.0, precision=precision, pcg_max_iters=3)
    got = capi.solve(mid, o)
    assert got[3]["cholesky_fallbacks"] == got[3]["iterations"] > 0
    chol = capi.solve(mid, capi.default_options(max_seconds=0.0, precision=precision, linear_solver=sfm.LINEAR_CHOLESKY))
    same(got, chol, atol=1e-10 if precision == 0 else 1e-7, cost_rtol=1e-12 if precision == 0 else 1e-9)
    same(got, mid_oracle, atol=1e-8 if precision == 0 else 2e-5, cost_rtol=1e-9 if precision == 0 else 1e-6)
    # resident problem: a starved solve followed by a normal one (the CG state must not be poisoned by the fallback)
    with capi.Problem(mid, precision=precision) as P:
        s1, _ = P.solve(o)
        P.reset()
        s2, _ = P.solve(capi.default_options(max_seconds=0.0, precision=precision))
        assert s1["cholesky_fallbacks"] > 0 and s2["cholesky_fallbacks"] == 0 and s2["linear_iters"] > 0
        assert abs(s1["final_cost"] - s2["final_cost"]) <= 1e-9 * s2["final_cost"]


def test_pcg_coarse_space_switch(capi, mid, mid_oracle):
    base = dict(max_seconds=0.0, linear_solver=1, pcg_tolerance=1e-10, pcg_anchored=0)
    on = capi.solve(mid, capi.default_options(**base))
    off = capi.solve(mid, capi.default_options(pcg_coarse_space=-1, **base))
    forced = capi.solve(mid, capi.default_options(pcg_coarse_space=1, **base))
    assert off[3]["linear_iters"] > 1.3 * on[3]["linear_iters"]          # plain block-Jacobi needs visibly more iterations
    assert forced[3]["linear_iters"] == on[3]["linear_iters"]
    for r in (on, off):
        same(r, mid_oracle, atol=1e-6)


def test_the_environment_no_longer_overrides_a_solve(capi, mid, monkeypatch):
    """ABI v5: no getenv below sfmba_problem_create*.  The ABI v4 override variables are ignored by a solve -- the
    fields of sfmba_options are the only way to a solver family -- and the slot that was pcg_persistent (the cooperative one-launch CG, never a
    default at any size, is gone; ABI v6: pcg_symmetric, a switch of the streaming CG) changes nothing at this size."""
    base = dict(max_seconds=0.0, linear_solver=1, pcg_tolerance=1e-10, pcg_anchored=0)
    on = capi.solve(mid, capi.default_options(**base))
    for var in ("SFMBA_PCG_COARSE", "SFMBA_PCG_SEGMENTS", "SFMBA_PCG_GATED", "SFMBA_PCG_ANCHOR", "SFMBA_EARLY_LINEARISE", "SFMBA_PCG_PERSISTENT"):
        monkeypatch.setenv(var, "0")
    env = capi.solve(mid, capi.default_options(pcg_symmetric=1, **base))
    assert env[3]["linear_iters"] == on[3]["linear_iters"] and env[3]["iterations"] == on[3]["iterations"]
    assert abs(env[3]["final_cost"] - on[3]["final_cost"]) <= 1e-12 * on[3]["final_cost"]
    off = capi.solve(mid, capi.default_options(pcg_coarse_space=-1, **base))
    assert off[3]["linear_iters"] > 1.3 * on[3]["linear_iters"]                # the field, and only the field, acts


def test_pcg_f32_matrix_switch(capi, sfm):
    prob = sfm.make_problem("cfg3", n_cam=230, n_pt=6000, seed=77)           # d = 1381: streaming CG
    ref = capi.solve(prob, capi.default_options(max_seconds=0.0, precision=1, linear_solver=0))
    for sw in (0, -1, 1):
        r = capi.solve(prob, capi.default_options(max_seconds=0.0, precision=1, linear_solver=1, pcg_f32_matrix=sw))
        same(r, ref, atol=2e-6)


def test_early_linearise_switch(capi, mid):
    a = capi.solve(mid, capi.default_options(max_seconds=0.0, precision=1, linear_solver=1))
    b = capi.solve(mid, capi.default_options(max_seconds=0.0, precision=1, linear_solver=1, early_linearise=-1))
    assert a[3]["iterations"] == b[3]["iterations"] and abs(a[3]["final_cost"] - b[3]["final_cost"]) <= 1e-9 * a[3]["final_cost"]
    assert np.abs(a[0] - b[0]).max() < 1e-6


def test_deterministic_create_flag(capi, sfm, mid):
    """SFMBA_CREATE_DETERMINISTIC through sfmba_problem_create_ex: bitwise reproducible across freshly built problems and re-solves;
    without the flag the handle is a normal one."""
    opt = capi.default_options(max_seconds=0.0, precision=1, linear_solver=1)
    runs = []
    for fresh in range(2):
        with capi.Problem(mid, precision=1, flags=sfm.CREATE_DETERMINISTIC) as P:
            for rep in range(2):
                P.reset()
                s, tr = P.solve(opt)
                runs.append((P.get_params(), s, [r["cost"] for r in tr]))
    for (cam, pt, f), s, costs in runs[1:]:
        assert np.array_equal(cam, runs[0][0][0]) and np.array_equal(pt, runs[0][0][1]) and f == runs[0][0][2]
        assert s["final_cost"] == runs[0][1]["final_cost"] and costs == runs[0][2]
    with pytest.raises(capi.SfmbaError):
        capi.Problem(mid, precision=1, flags=64)              # unknown flag: refused


def test_auto_takes_the_same_path_on_every_solve_of_a_resident_problem(capi, sfm):
    """AUTO's "this structure prefers the factorisation" used to be remembered on the problem, so solve / reset / solve took CG
    then Cholesky the first time and Cholesky only the second -- results ~1e-10 apart, against the bitwise guarantee of
    SFMBA_CREATE_DETERMINISTIC.  The switch is local to a solve now.  A banded reduced system makes AUTO switch mid-solve."""
    prob = sfm.make_problem("cfg3_banded", n_cam=60, n_pt=8000)
    opt = capi.default_options(max_seconds=0.0, precision=1)           # the library default: SFMBA_LINEAR_AUTO
    runs = []
    with capi.Problem(prob, precision=1, flags=sfm.CREATE_DETERMINISTIC) as P:
        for rep in range(3):
            P.reset()
            s, tr = P.solve(opt)
            runs.append((P.get_params(), s, [r["cost"] for r in tr], [r["linear_iters"] for r in tr]))
    assert runs[0][1]["termination_name"] == "CONVERGENCE"
    for (cam, pt, f), s, costs, lin in runs[1:]:
        assert lin == runs[0][3]                                          # the same solver path, iteration by iteration
        assert np.array_equal(cam, runs[0][0][0]) and np.array_equal(pt, runs[0][0][1]) and f == runs[0][0][2]
        assert s["final_cost"] == runs[0][1]["final_cost"] and costs == runs[0][2]


def test_sharded_switches_and_deterministic_sharded(capi, sfm, mid):
    """One rank of the native sharded loop (the collectives are no-ops; pack / unpack and the kernels are not): two-phase exchange off,
    deterministic accumulation in the sharded path (ABI v4: was excluded), each against the unsharded solve."""
    from sfm_toy_library_amd.sharded import HipShardBackend, solve_sharded_native
    opt = capi.default_options(max_seconds=0.0, linear_solver=1, precision=1)
    ref = capi.solve(mid, opt)
    be = HipShardBackend(mid, 0, 1, device=0, precision=1)
    try:
        for okw in (dict(), dict(shard_two_phase=-1), dict(early_linearise=-1), dict(pcg_coarse_space=-1), dict(shard_distributed_cg=1),
                    dict(shard_distributed_cg=1, pcg_coarse_space=-1), dict(shard_distributed_cg=2), dict(shard_distributed_cg=2, pcg_coarse_space=-1)):
            be.reset()
            s = solve_sharded_native(be, capi.default_options(max_seconds=0.0, linear_solver=1, precision=1, **okw), comm=None)
            assert s["termination_name"] == "CONVERGENCE" and s["iterations"] == ref[3]["iterations"]
            assert abs(s["final_cost"] - ref[3]["final_cost"]) <= 1e-9 * ref[3]["final_cost"]
            assert np.abs(be.get_params()[0] - ref[0]).max() < (5e-6 if okw.get("shard_distributed_cg", 0) == 2 else 2e-6)     # (F32J: the implicit product rounds elsewhere)
            assert s["distributed_cg"] == bool(okw.get("shard_distributed_cg", 0) >= 1) and s["implicit_schur_cg"] == bool(okw.get("shard_distributed_cg", 0) == 2)
    finally:
        be.close()
    runs = []
    for fresh in range(2):
        be = HipShardBackend(mid, 0, 1, device=0, precision=1, flags=sfm.CREATE_DETERMINISTIC)
        try:
            for rep in range(2):
                be.reset()
                s = solve_sharded_native(be, opt, comm=None)
                runs.append((be.get_params(), s["final_cost"]))
        finally:
            be.close()
    for (cam, pt, f), cost in runs[1:]:
        assert np.array_equal(cam, runs[0][0][0]) and np.array_equal(pt, runs[0][0][1]) and f == runs[0][0][2] and cost == runs[0][1]
    assert abs(runs[0][1] - ref[3]["final_cost"]) <= 1e-9 * ref[3]["final_cost"] and np.abs(runs[0][0][0] - ref[0]).max() < 2e-6


def test_implicit_schur_cg_with_duplicate_observations(capi, sfm, mid):
    """shard_distributed_cg = 2 applies the reduced matrix implicitly, from ALL pairs of observations of a point -- pairs of ONE camera on a
    point (duplicate observations, which the C ABI allows) included.  Round 4 made such a problem step aside to form 1, decided from the
    rank's OWN duplicate count: ranks could disagree on the form and issue different collectives.  Now the duplicates' cross
    terms are simply not added to the diagonal blocks in this form (they stay a preconditioner) and the implicit product carries them:
    every rank takes the form the options name, and the result is the unsharded solve's."""
    from sfm_toy_library_amd.sharded import HipShardBackend, solve_sharded_native
    extra = np.arange(0, mid.n_obs, 7)
    dup = sfm.BAProblem(mid.cam6, mid.pt3, mid.focal, np.concatenate([mid.obs_cam, mid.obs_cam[extra]]),
                        np.concatenate([mid.obs_pt, mid.obs_pt[extra]]), np.concatenate([mid.obs_xy, mid.obs_xy[extra] + 0.25]))
    opt = capi.default_options(max_seconds=0.0, linear_solver=1, precision=0, pcg_tolerance=1e-12, pcg_anchored=0)
    for prob in (dup, mid):
        ref = capi.solve(prob, opt)
        be = HipShardBackend(prob, 0, 1, device=0, precision=0)
        try:
            s = solve_sharded_native(be, capi.default_options(max_seconds=0.0, linear_solver=1, precision=0, pcg_tolerance=1e-12, pcg_anchored=0,
                                                              shard_distributed_cg=2), comm=None)
            assert s["distributed_cg"] and s["implicit_schur_cg"] and s["termination_name"] == "CONVERGENCE"
            assert s["iterations"] == ref[3]["iterations"] and abs(s["final_cost"] - ref[3]["final_cost"]) <= 1e-9 * ref[3]["final_cost"]
            assert np.abs(be.get_params()[0] - ref[0]).max() < 1e-7
        finally:
            be.close()


def test_poisoned_handle_contract_and_append_argument_errors(capi, sfm):
    """Argument errors of sfmba_problem_append are detected before anything is touched: the handle stays usable."""
    prob = sfm.make_problem("tiny")
    with capi.Problem(prob) as P:
        with pytest.raises(capi.SfmbaError):
            P.append(prob.cam6, prob.pt3, prob.focal, np.array([99], np.int32), np.array([0], np.int32), np.zeros((1, 2)))
        s, _ = P.solve(capi.default_options(max_seconds=0.0))
        assert s["termination_name"] == "CONVERGENCE"
