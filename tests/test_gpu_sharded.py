"""Sharded solve through the HIP C ABI (-m gpu): two ranks (two processes) share the one MI355X of the
test box, each holds half of the points, the reduced camera system is all-reduced between them (gloo
on device tensors here; RCCL when every rank has its own GPU).  Must reproduce the ORACLE's solve of the
whole problem (and, as a consistency check, the single-GPU HIP solve)."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


CASES = {
    # name: (make_problem kwargs, precision, linear solver, options)
    "small": (dict(name="small"), 0, 0, dict()),
    "cfg2": (dict(name="cfg2"), 0, 1, dict(pcg_tolerance=1e-12, pcg_anchored=0)),
    # 230 cameras: reduced dimension 1381 > 1280 -> streaming CG (fp32-stored matrix in F32J mode); ~10 pairs per 6x6
    # block -> the sixteen-lane pair pass k_schur_pairs_sub_f; sharded ranks transform the all-reduced system (k_pcg_transform)
    "wide": (dict(name="cfg3", n_cam=230, n_pt=6000, seed=77), 1, 1, dict()),
    "wide_x64": (dict(name="cfg3", n_cam=230, n_pt=6000, seed=77), 1, 1, dict(shard_f32_exchange=-1)),       # every exchange fp64
    # point counts that no world size of 2, 3 or 4 divides (5003 is prime, 6001 = 17 * 353): ranks hold shards of different sizes
    "cfg2_uneven": (dict(name="cfg2", n_pt=5003), 0, 1, dict(pcg_tolerance=1e-12, pcg_anchored=0)),
    "wide_uneven": (dict(name="cfg3", n_cam=230, n_pt=6001, seed=78), 1, 1, dict()),
    # the library default (AUTO: CG to 1e-12; d = 361 > 256) and the exact solver on an uneven split
    "auto_uneven": (dict(name="cfg3", n_cam=60, n_pt=8003, seed=5), 0, 2, dict()),
    "chol_uneven": (dict(name="cfg2", n_pt=5003), 0, 0, dict()),
    # the CG without the redundant solve: reduce-scatter of the blocks, products from the owned blocks, one small
    # all-reduce per CG iteration -- fp64 blocks, fp32 blocks (d = 1381 > 1280, F32J), the library default, plain block-Jacobi
    "dist_cfg2": (dict(name="cfg2", n_pt=5003), 0, 1, dict(pcg_tolerance=1e-12, pcg_anchored=0, shard_distributed_cg=1)),
    "dist_wide": (dict(name="cfg3", n_cam=230, n_pt=6001, seed=78), 1, 1, dict(shard_distributed_cg=1)),
    "dist_auto": (dict(name="cfg3", n_cam=60, n_pt=8003, seed=5), 0, 2, dict(shard_distributed_cg=1)),
    "dist_plain": (dict(name="cfg3", n_cam=60, n_pt=8003, seed=5), 1, 1, dict(shard_distributed_cg=1, pcg_coarse_space=-1)),
    # ... and WITHOUT any exchange of the reduced matrix: the CG product formed implicitly from every rank's own points
    # (ba_kernels.hip "Implicit Schur product"), one all-reduce of ld doubles per CG iteration and nothing else -- fp64, fp32 Jacobian
    # blocks at d = 1381, the library default (AUTO), plain block-Jacobi
    "dist_imp_cfg2": (dict(name="cfg2", n_pt=5003), 0, 1, dict(pcg_tolerance=1e-12, pcg_anchored=0, shard_distributed_cg=2)),
    "dist_imp_wide": (dict(name="cfg3", n_cam=230, n_pt=6001, seed=78), 1, 1, dict(shard_distributed_cg=2)),
    "dist_imp_auto": (dict(name="cfg3", n_cam=60, n_pt=8003, seed=5), 0, 2, dict(shard_distributed_cg=2)),
    "dist_imp_plain": (dict(name="cfg3", n_cam=60, n_pt=8003, seed=5), 1, 1, dict(shard_distributed_cg=2, pcg_coarse_space=-1)),
    # ... and with the block ROWS of the reduced matrix sharded (SFMBA_CREATE_ROW_SHARDED: every rank holds the whole problem,
    # the per-point table is all-gathered, the pair pass forms the rank's own block rows from all their pairs, the multi-workgroup distributed CG
    # runs on them) -- fp64 blocks, fp32 blocks (d = 1381, F32J), the library default (AUTO), plain block-Jacobi, the reference's solver choice
    "row_cfg2": (dict(name="cfg2", n_pt=5003), 0, 1, dict(pcg_tolerance=1e-12, pcg_anchored=0)),
    "row_wide": (dict(name="cfg3", n_cam=230, n_pt=6001, seed=78), 1, 1, dict()),
    "row_auto": (dict(name="cfg3", n_cam=60, n_pt=8003, seed=5), 0, 2, dict()),
    "row_plain": (dict(name="cfg3", n_cam=60, n_pt=8003, seed=5), 1, 1, dict(pcg_coarse_space=-1)),
    "row_chol": (dict(name="cfg2", n_pt=5003), 0, 0, dict()),
    # edge cases of the ownership: more ranks than block rows (two cameras = one off-diagonal block: ranks without a row, without a pair pass),
    # a handful of cameras (d = 43: the one-launch Cholesky size, served by the CG here), fewer points than a rank's stride would suggest
    "row_two_cams": (dict(name="cfg2", n_cam=2, n_pt=301, views=2, seed=5), 0, 1, dict(pcg_tolerance=1e-12, pcg_anchored=0)),
    "row_tiny": (dict(name="tiny"), 0, 2, dict()),
    # degenerate shapes through every sharded form (test_sharded_edge_shapes): one view (d = 7, no off-diagonal block at all), tracks of length one
    # (no pair anywhere: S is block diagonal + the focal border), a non-finite observation (FAILURE on every rank, nothing moves)
    "row_one_cam": (dict(name="cfg2", n_cam=1, n_pt=201, views=1, seed=101), 0, 2, dict()),
    "row_views1": (dict(name="cfg2", n_cam=5, n_pt=301, views=1, seed=105), 0, 1, dict(pcg_tolerance=1e-12, pcg_anchored=0)),
    "row_nan": (dict(name="tiny"), 0, 2, dict()),
    "rep_views1": (dict(name="cfg2", n_cam=5, n_pt=301, views=1, seed=105), 0, 2, dict()),
    "dist_views1": (dict(name="cfg2", n_cam=5, n_pt=301, views=1, seed=105), 0, 1, dict(pcg_tolerance=1e-12, pcg_anchored=0, shard_distributed_cg=1)),
    "dist_imp_one_cam": (dict(name="cfg2", n_cam=1, n_pt=201, views=1, seed=101), 0, 2, dict(shard_distributed_cg=2)),
    "dist_nan": (dict(name="tiny"), 0, 2, dict(shard_distributed_cg=1)),
}


def _edge_problem(sfm, case):
    kw = CASES[case][0]
    prob = sfm.make_problem(**kw)
    if case.endswith("_nan"):
        prob.obs_xy[7, 1] = np.nan
    return prob


def _worker(rank, world, port, case, out, native=False, n_repeats=12, flags=0):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sfm_toy_library_amd as sfm
    from sfm_toy_library_amd import capi
    from sfm_toy_library_amd.sharded import HipShardBackend, HipRowShardBackend, solve_sharded, solve_sharded_native
    kw, precision, linear, okw = CASES[case]
    prob = _edge_problem(sfm, case)
    backend = (HipRowShardBackend if case.startswith("row_") else HipShardBackend)(prob, rank, world, device=0, precision=precision, flags=flags)
    opt = capi.default_options(max_seconds=0.0, precision=precision, linear_solver=linear, **okw)
    summ = solve_sharded_native(backend, opt, dist=dist) if native else solve_sharded(backend, dist, opt)
    cam, pt, f = backend.get_params()
    # native loop: the same solve again and again.  Two processes on one GPU delay each other's workgroups, which is what
    # exposes a launch acting on a flag raised by its own first workgroup (the CG's early exit once did: replicas drifted apart).
    repeats = []
    for _ in range(n_repeats if native else 0):
        backend.reset()
        solve_sharded_native(backend, opt, dist=dist)
        repeats.append(backend.get_params()[0].copy())
    out.put((rank, summ, cam, pt, f, backend._point_range, repeats))
    dist.barrier()
    backend.close()
    dist.destroy_process_group()


def _run_ranks(world, port, case, native, n_repeats, flags=0, timeout=180):
    """Spawns the ranks, collects their results; whatever happens, no rank outlives the test (a rank stuck in a collective whose peer died
    would keep the test session from exiting).  Several processes on ONE GPU with a gloo rendezvous is a test-only arrangement (the product
    is one process per GPU): a run in which no rank reports within the timeout is repeated ONCE on another port, with a warning in the
    session summary -- a wrong result is never retried."""
    import queue, warnings
    ctx = mp.get_context("spawn")
    for attempt in range(2):
        out = ctx.Queue()
        procs = [ctx.Process(target=_worker, args=(r, world, port + 1000 * attempt, case, out, native, n_repeats, flags)) for r in range(world)]
        for p in procs:
            p.start()
        try:
            results = sorted([out.get(timeout=timeout) for _ in range(world)], key=lambda t: t[0])
            for p in procs:
                p.join(timeout=60)
                assert p.exitcode == 0
            return results
        except queue.Empty:
            if attempt == 1:
                raise
            warnings.warn("sharded test %s (world %d): no result within %d s, ranks killed and the run repeated once" % (case, world, timeout))
        finally:
            for p in procs:
                if p.is_alive():
                    p.kill()
                    p.join(timeout=10)


@pytest.mark.parametrize("case,native,x32", [("small", False, True), ("cfg2", False, True), ("wide", False, True), ("cfg2", True, True),
                                             ("wide", True, True), ("wide", True, False)])
def test_two_rank_sharded_hip_solve(sfm, oracle, monkeypatch, case, native, x32):
    """native: the LM loop runs inside the C library (sfmba_problem_solve_sharded), the collectives come back through a callback.
    x32 (only the 'wide' native case has the fp32-stored matrix): the off-diagonal blocks of the preconditioned matrix are exchanged in
    single precision (sfmba_problem_set_allreduce_f32); False switches that off (every exchange fp64)."""
    from sfm_toy_library_amd import capi
    kw, precision, linear, okw = CASES["wide_x64" if not x32 else case]
    world, port = 2, 29711 + (os.getpid() % 500)
    results = _run_ranks(world, port, "wide_x64" if not x32 else case, native, 12)
    prob = sfm.make_problem(**kw)
    (r0, s0, cam0, pt0, f0, rng0, rep0), (r1, s1, cam1, pt1, f1, rng1, rep1) = results
    for a, b in zip(rep0, rep1):
        assert np.array_equal(a, b)                                     # every repeat: replicas bit-identical
        assert np.abs(a - cam0).max() < 1e-9                            # and the same solve (atomics reorder the last bits)
    pts = np.vstack([pt0, pt1])
    assert rng0 == (0, prob.n_pt // 2) and rng1[1] == prob.n_pt
    assert s0["final_cost"] == s1["final_cost"]
    if native:
        wide_f32 = case == "wide" and x32           # the only case with the fp32-stored matrix (d = 1381 > 1280, F32J)
        assert s0["exchange_b_fp32"] == wide_f32 and s0["exchange_bytes"][1] == (4 if wide_f32 else 8) * 18 * prob.n_cam * (prob.n_cam - 1)
    assert np.array_equal(cam0, cam1) and f0 == f1                      # replicas stay bit-identical
    # --- against the oracle's solve of the WHOLE problem (Ceres-equivalent LM + DENSE_SCHUR on one host) ---
    cam_o, pt_o, f_o, s_o, _ = oracle.solve(prob, sfm.SfmbaOptions.defaults(max_seconds=0.0))
    exact = precision == 0
    assert s0["termination_name"] == s1["termination_name"] == s_o["termination_name"] == "CONVERGENCE"
    assert s0["iterations"] == s1["iterations"] == s_o["iterations"]
    assert np.isclose(s0["initial_cost"], s_o["initial_cost"], rtol=1e-12 if exact else 1e-9)
    assert abs(s0["final_cost"] - s_o["final_cost"]) <= (1e-9 if exact else 1e-6) * s_o["final_cost"]
    assert abs(np.sqrt(2 * s0["final_cost"] / prob.n_obs) - np.sqrt(2 * s_o["final_cost"] / prob.n_obs)) < 1e-4
    atol = 1e-7 if exact else 5e-6
    assert np.abs(cam0 - cam_o).max() <= atol and np.isclose(f0, f_o, rtol=1e-9 if exact else 1e-7)
    assert np.abs(pts - pt_o).max() <= atol
    # --- and against the unsharded HIP solve with the same options ---
    cam_s, pt_s, f_s, s_s, _ = capi.solve(prob, capi.default_options(max_seconds=0.0, precision=precision, linear_solver=linear, **okw))
    assert s0["iterations"] == s_s["iterations"]
    assert abs(s0["final_cost"] - s_s["final_cost"]) <= (1e-9 if exact else 1e-7) * s_s["final_cost"]
    assert np.allclose(cam0, cam_s, atol=1e-8 if exact else 5e-6) and np.allclose(pts, pt_s, atol=1e-8 if exact else 5e-6)


def test_one_rank_rccl_communicator_and_native_loop(sfm, oracle):
    """The box has one GPU: an RCCL communicator of ONE rank still goes through ncclCommInitRank / ncclAllReduce on the solver's
    stream (sfmba_comm_*).  Result = the oracle's; the native loop without any collective (world 1, no communicator) as well."""
    import torch.distributed as dist
    from sfm_toy_library_amd import capi
    from sfm_toy_library_amd.sharded import HipShardBackend, RcclComm, solve_sharded_native
    prob = sfm.make_problem("cfg2")
    want = oracle.solve(prob, sfm.SfmbaOptions.defaults(max_seconds=0.0))
    be = HipShardBackend(prob, 0, 1, device=0, precision=0)
    opt = capi.default_options(max_seconds=0.0, linear_solver=1)
    comm = RcclComm(None, 0, 1, device=0)
    try:
        for c in (comm, None):
            be.reset()
            s = solve_sharded_native(be, opt, comm=c)
            cam, pt, f = be.get_params()
            assert s["termination_name"] == "CONVERGENCE" and s["iterations"] == want[3]["iterations"]
            assert abs(s["final_cost"] - want[3]["final_cost"]) <= 1e-9 * want[3]["final_cost"]
            assert np.abs(cam - want[0]).max() < 1e-6 and np.abs(pt - want[1]).max() < 1e-6
    finally:
        comm.close(); be.close()


@pytest.mark.parametrize("linear", [0, 1])
def test_sharded_exchange_variants_agree(sfm, monkeypatch, linear):
    """CG path: two all-reduces per linearisation (diagonal blocks + vectors, then the preconditioned off-diagonal blocks) against
    the single all-reduce of the whole system followed by the transform (SFMBA_SHARD_TWO_PHASE=0); and both against the unsharded
    solve.  linear = 0: the exact solver (always the single all-reduce) against the unsharded exact solve."""
    from sfm_toy_library_amd import capi
    from sfm_toy_library_amd.sharded import HipShardBackend, solve_sharded_native
    prob = sfm.make_problem("cfg3", n_cam=60, n_pt=8000, seed=5)
    opt = capi.default_options(max_seconds=0.0, linear_solver=linear, precision=1)
    ref_cam, ref_pt, _, ref, _ = capi.solve(prob, opt)
    be = HipShardBackend(prob, 0, 1, device=0, precision=1)
    try:
        got = {}
        for flag in ("1", "0"):
            be.reset()
            s = solve_sharded_native(be, capi.default_options(max_seconds=0.0, linear_solver=linear, precision=1, shard_two_phase=1 if flag == "1" else -1), comm=None)
            got[flag] = (s, be.get_params())
            assert s["termination_name"] == ref["termination_name"] and s["iterations"] == ref["iterations"]
            assert abs(s["final_cost"] - ref["final_cost"]) <= 1e-9 * ref["final_cost"]
            assert np.abs(got[flag][1][0] - ref_cam).max() < 2e-6 and np.abs(got[flag][1][1] - ref_pt).max() < 2e-6
        assert np.abs(got["1"][1][0] - got["0"][1][0]).max() < 2e-6
    finally:
        be.close()


@pytest.mark.parametrize("world,case", [(2, "cfg2_uneven"), (3, "cfg2_uneven"), (4, "cfg2_uneven"), (3, "wide_uneven"), (4, "wide_uneven"),
                                        (3, "auto_uneven"), (4, "chol_uneven"),
                                        (2, "dist_cfg2"), (3, "dist_cfg2"), (4, "dist_wide"), (3, "dist_auto"), (2, "dist_plain"),
                                        (2, "dist_imp_cfg2"), (3, "dist_imp_cfg2"), (4, "dist_imp_wide"), (3, "dist_imp_auto"), (2, "dist_imp_plain"),
                                        (4, "dist_imp_cfg2")])
def test_multi_rank_uneven_sharded_hip_solve(sfm, oracle, world, case):
    """2, 3 and 4 ranks (processes) on the one MI355X of the box, point counts the world size does not divide
    (shards of different sizes), native C loop with the collectives through the callback -- against the ORACLE's solve of the whole
    problem; replicas must stay bit-identical."""
    from sfm_toy_library_amd import capi
    kw, precision, linear, okw = CASES[case]
    port = 29311 + (os.getpid() % 300) + 7 * world
    results = _run_ranks(world, port, case, True, 2)
    prob = sfm.make_problem(**kw)
    assert prob.n_pt % world != 0
    ranges = [r[5] for r in results]
    assert ranges[0][0] == 0 and ranges[-1][1] == prob.n_pt and all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
    assert len({hi - lo for lo, hi in ranges}) > 1                       # genuinely uneven
    cam0, f0, s0 = results[0][2], results[0][4], results[0][1]
    assert s0["distributed_cg"] == case.startswith("dist_") and s0["implicit_schur_cg"] == case.startswith("dist_imp_")
    if case.startswith("dist_imp_"):
        assert s0["exchange_bytes"][1] == 0                              # nothing of the reduced matrix crosses the ranks
    for r in results[1:]:
        assert np.array_equal(r[2], cam0) and r[4] == f0 and r[1]["final_cost"] == s0["final_cost"]     # replicas bit-identical
        for a, b in zip(results[0][6], r[6]):
            assert np.array_equal(a, b)
    pts = np.vstack([r[3] for r in results])
    cam_o, pt_o, f_o, s_o, _ = oracle.solve(prob, sfm.SfmbaOptions.defaults(max_seconds=0.0))
    exact = precision == 0
    assert s0["termination_name"] == s_o["termination_name"] == "CONVERGENCE" and s0["iterations"] == s_o["iterations"]
    assert abs(s0["final_cost"] - s_o["final_cost"]) <= (1e-9 if exact else 1e-6) * s_o["final_cost"]
    assert abs(np.sqrt(2 * s0["final_cost"] / prob.n_obs) - np.sqrt(2 * s_o["final_cost"] / prob.n_obs)) < 1e-4
    atol = 1e-7 if exact else 5e-6
    assert np.abs(cam0 - cam_o).max() <= atol and np.isclose(f0, f_o, rtol=1e-9 if exact else 1e-7)
    assert np.abs(pts - pt_o).max() <= atol


@pytest.mark.parametrize("world,case,flags", [(2, "row_cfg2", 0), (3, "row_cfg2", 0), (4, "row_wide", 0), (3, "row_auto", 0), (2, "row_plain", 0),
                                              (4, "row_cfg2", 0), (3, "row_wide", 0), (2, "row_chol", 0), (3, "row_cfg2", 1),
                                              (3, "row_two_cams", 0), (4, "row_tiny", 0)])
def test_row_sharded_hip_solve(sfm, oracle, world, case, flags):
    """block rows of the reduced matrix per rank (options.shard_distributed_cg = 3, SFMBA_CREATE_ROW_SHARDED).  2, 3 and 4
    ranks (processes) on the one MI355X, point counts no world size divides, native C loop with the collectives -- all-reduce AND the
    all-gather of the per-point table -- through callbacks; against the ORACLE's solve of the whole problem.  Every rank ends up with the
    WHOLE solution (the final points are all-gathered): replicas bit-identical in cameras AND points.  flags = 1: deterministic handles."""
    kw, precision, linear, okw = CASES[case]
    port = 29011 + (os.getpid() % 300) + 11 * world
    results = _run_ranks(world, port, case, True, 2, flags)
    prob = sfm.make_problem(**kw)
    assert prob.n_pt % world != 0 or case == "row_tiny"
    cam0, pts0, f0, s0 = results[0][2], results[0][3], results[0][4], results[0][1]
    assert s0["distributed_cg"] and s0["row_sharded"] and not s0["implicit_schur_cg"]
    assert pts0.shape == (prob.n_pt, 3)
    for r in results[1:]:
        assert np.array_equal(r[2], cam0) and r[4] == f0 and r[1]["final_cost"] == s0["final_cost"]     # replicas bit-identical
        assert np.array_equal(r[3], pts0)                                                               # ... points included
        for a, b in zip(results[0][6], r[6]):
            assert np.array_equal(a, b)
    if flags & 1:                                            # deterministic handles: the repeats are the first solve bit for bit
        for a in results[0][6]:
            assert np.array_equal(a, cam0)
    cam_o, pt_o, f_o, s_o, _ = oracle.solve(prob, sfm.SfmbaOptions.defaults(max_seconds=0.0))
    exact = precision == 0
    assert s0["termination_name"] == s_o["termination_name"] == "CONVERGENCE" and s0["iterations"] == s_o["iterations"]
    assert abs(s0["final_cost"] - s_o["final_cost"]) <= (1e-9 if exact else 1e-6) * s_o["final_cost"]
    assert abs(np.sqrt(2 * s0["final_cost"] / prob.n_obs) - np.sqrt(2 * s_o["final_cost"] / prob.n_obs)) < 1e-4
    atol = 1e-7 if exact else 5e-6
    assert np.abs(cam0 - cam_o).max() <= atol and np.isclose(f0, f_o, rtol=1e-9 if exact else 1e-7)
    assert np.abs(pts0 - pt_o).max() <= atol


@pytest.mark.parametrize("world,case", [(2, "row_one_cam"), (3, "row_views1"), (2, "row_nan"), (2, "rep_views1"), (3, "dist_views1"),
                                        (2, "dist_imp_one_cam"), (3, "dist_nan")])
def test_sharded_edge_shapes(sfm, oracle, world, case):
    """The degenerate inputs of test_gpu_edge_cases.py through the sharded entry points (native loop, gloo callbacks, several ranks on the one GPU):
    ranks that own no block row / no pair / no point of a camera, and the FAILURE exit taken by every rank from the same all-reduced cost."""
    kw, precision, linear, okw = CASES[case]
    results = _run_ranks(world, 29411 + (os.getpid() % 300) + 13 * world, case, True, 1)
    prob = _edge_problem(sfm, case)
    cam_o, pt_o, f_o, s_o, tr_o = oracle.solve(prob, sfm.SfmbaOptions.defaults(max_seconds=0.0))
    s0, cam0, f0 = results[0][1], results[0][2], results[0][4]
    pts = results[0][3] if case.startswith("row_") else np.vstack([r[3] for r in results])
    for r in results[1:]:
        assert r[1]["termination_name"] == s0["termination_name"] and r[1]["iterations"] == s0["iterations"]
        assert np.array_equal(r[2], cam0, equal_nan=True) and r[4] == f0
    assert s0["termination_name"] == s_o["termination_name"] and s0["iterations"] == s_o["iterations"]
    if case.endswith("_nan"):
        assert s0["termination_name"] == "FAILURE" and np.array_equal(cam0, prob.cam6) and np.array_equal(pts, prob.pt3) and f0 == prob.focal
        return
    assert abs(s0["final_cost"] - s_o["final_cost"]) <= 1e-9 * s_o["final_cost"] + 1e-14 * s_o["initial_cost"]
    assert np.abs(cam0 - cam_o).max() < 1e-6 and np.abs(pts - pt_o).max() < 1e-6 and abs(f0 - f_o) < 1e-6 * abs(f_o)


@pytest.mark.parametrize("case", ["row_cfg2", "row_wide", "row_auto"])
def test_row_sharded_one_rank_with_and_without_rccl(sfm, oracle, case):
    """One rank: the row-sharded loop through an RCCL communicator of one rank (ncclAllReduce AND ncclAllGather on the solver's stream)
    and without any collective call; both = the oracle's solve."""
    from sfm_toy_library_amd import capi
    from sfm_toy_library_amd.sharded import HipRowShardBackend, RcclComm, solve_sharded_native
    kw, precision, linear, okw = CASES[case]
    prob = sfm.make_problem(**kw)
    want = oracle.solve(prob, sfm.SfmbaOptions.defaults(max_seconds=0.0))
    be = HipRowShardBackend(prob, 0, 1, device=0, precision=precision)
    opt = capi.default_options(max_seconds=0.0, precision=precision, linear_solver=linear, **okw)
    comm = RcclComm(None, 0, 1, device=0)
    exact = precision == 0
    try:
        for c in (comm, None):
            be.reset()
            s = solve_sharded_native(be, opt, comm=c)
            cam, pt, f = be.get_params()
            assert s["row_sharded"] and s["termination_name"] == "CONVERGENCE" and s["iterations"] == want[3]["iterations"]
            assert abs(s["final_cost"] - want[3]["final_cost"]) <= (1e-9 if exact else 1e-6) * want[3]["final_cost"]
            assert np.abs(cam - want[0]).max() < (1e-7 if exact else 5e-6) and np.abs(pt - want[1]).max() < (1e-7 if exact else 5e-6)
    finally:
        comm.close(); be.close()


def test_rccl_bindings_execute_on_one_rank():
    """sfmba_comm_reduce_scatter, sfmba_comm_allreduce_f32 and sfmba_comm_allgather had never been executed (the 2 - 4
    rank tests go through gloo callbacks).  A communicator of ONE rank drives each binding through RCCL itself; with one rank every
    collective is the identity on its input."""
    import ctypes as C
    import torch
    from sfm_toy_library_amd import capi
    from sfm_toy_library_amd.sharded import RcclComm
    L = capi.lib()
    comm = RcclComm(None, 0, 1, device=0)
    try:
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        g = torch.Generator(device="cuda").manual_seed(3)
        a64 = torch.rand(4099, dtype=torch.float64, device="cuda", generator=g); w64 = a64.clone()
        assert L.sfmba_comm_allreduce(comm._h, C.c_void_p(a64.data_ptr()), C.c_int64(a64.numel()), st) == 0
        a32 = torch.rand(4099, dtype=torch.float32, device="cuda", generator=g); w32 = a32.clone()
        assert L.sfmba_comm_allreduce_f32(comm._h, C.c_void_p(a32.data_ptr()), C.c_int64(a32.numel()), st) == 0
        r64 = torch.rand(3 * 36 * 7, dtype=torch.float64, device="cuda", generator=g); wr64 = r64.clone()
        assert L.sfmba_comm_reduce_scatter(comm._h, C.c_void_p(r64.data_ptr()), C.c_void_p(r64.data_ptr()), C.c_int64(r64.numel()), C.c_int(0), st) == 0
        r32 = torch.rand(3 * 36 * 7, dtype=torch.float32, device="cuda", generator=g); wr32 = r32.clone()
        assert L.sfmba_comm_reduce_scatter(comm._h, C.c_void_p(r32.data_ptr()), C.c_void_p(r32.data_ptr()), C.c_int64(r32.numel()), C.c_int(1), st) == 0
        gb = torch.randint(0, 255, (64 * 1001,), dtype=torch.uint8, device="cuda", generator=g); wgb = gb.clone()
        assert L.sfmba_comm_allgather(comm._h, C.c_void_p(gb.data_ptr()), C.c_int64(gb.numel()), st) == 0
        torch.cuda.synchronize()
        assert torch.equal(a64, w64) and torch.equal(a32, w32) and torch.equal(r64, wr64) and torch.equal(r32, wr32) and torch.equal(gb, wgb)
    finally:
        comm.close()
