"""Sharded solve through the HIP C ABI (-m gpu): two ranks (two processes) share the one MI355X of the
test box, each holds half of the points, the reduced camera system is all-reduced between them (gloo
on device tensors here; RCCL when every rank has its own GPU).  Must reproduce the single-GPU solve."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, name, linear, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sfm_toy_library_amd as sfm
    from sfm_toy_library_amd import capi
    from sfm_toy_library_amd.sharded import HipShardBackend, solve_sharded
    prob = sfm.make_problem(name)
    backend = HipShardBackend(prob, rank, world, device=0, precision=0)
    opt = capi.default_options(max_seconds=0.0, linear_solver=linear, pcg_tolerance=1e-12, pcg_anchored=0)
    summ = solve_sharded(backend, dist, opt)
    cam, pt, f = backend.get_params()
    out.put((rank, summ, cam, pt, f, backend._point_range))
    dist.barrier()
    backend.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,linear", [("small", 0), ("cfg2", 1)])
def test_two_rank_sharded_hip_solve(sfm, name, linear):
    from sfm_toy_library_amd import capi
    world, port = 2, 29711 + (os.getpid() % 500)
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, linear, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([out.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    prob = sfm.make_problem(name)
    cam_s, pt_s, f_s, s_s, _ = capi.solve(prob, capi.default_options(max_seconds=0.0, linear_solver=linear, pcg_tolerance=1e-12, pcg_anchored=0))
    (r0, s0, cam0, pt0, f0, rng0), (r1, s1, cam1, pt1, f1, rng1) = results
    assert s0["termination_name"] == s1["termination_name"] == s_s["termination_name"] == "CONVERGENCE"
    assert s0["iterations"] == s1["iterations"] == s_s["iterations"]
    assert np.isclose(s0["initial_cost"], s_s["initial_cost"], rtol=1e-12)
    assert abs(s0["final_cost"] - s_s["final_cost"]) <= 1e-9 * s_s["final_cost"]
    assert s0["final_cost"] == s1["final_cost"]
    assert np.array_equal(cam0, cam1) and f0 == f1                      # replicas stay bit-identical
    assert np.allclose(cam0, cam_s, atol=1e-8) and np.isclose(f0, f_s, rtol=1e-10)
    pts = np.vstack([pt0, pt1])
    assert rng0 == (0, prob.n_pt // 2) and rng1[1] == prob.n_pt
    assert np.allclose(pts, pt_s, atol=1e-8)
