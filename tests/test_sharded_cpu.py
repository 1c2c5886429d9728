"""world_size-2 gloo test of the sharded LM choreography (sfm_toy_library_amd.sharded.solve_sharded):
two processes, each holding half of the points, exchange only the three all-reduce buffers and must
reproduce the single-process oracle (same iterations, same final cost)."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sfm_toy_library_amd as sfm
    from sfm_toy_library_amd.sharded import solve_sharded
    from shard_cpu_backend import CpuShardBackend
    prob = sfm.make_problem("tiny")
    backend = CpuShardBackend(prob, rank, world)
    summ = solve_sharded(backend, dist, sfm.SfmbaOptions.defaults(max_seconds=0.0))
    out.put((rank, summ, backend.cam.copy(), backend.f))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_solve_matches_oracle(oracle, sfm):
    import queue
    import socket
    world = 2
    ctx = mp.get_context("spawn")
    results = None
    for attempt in range(2):           # (a rendezvous that does not come up -- the port taken between probing and binding -- is repeated once; a result never is)
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        out = ctx.Queue()
        procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
        for p in procs:
            p.start()
        try:
            results = sorted([out.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
        except queue.Empty:
            results = None
        for p in procs:
            p.join(timeout=60 if results is not None else 1)
            if p.is_alive():
                p.kill()
                p.join()
        if results is not None:
            assert all(p.exitcode == 0 for p in procs)
            break
    assert results is not None, "no rank reported in two attempts"
    prob = sfm.make_problem("tiny")
    cam_o, pt_o, f_o, s_o, _ = oracle.solve(prob, sfm.SfmbaOptions.defaults(max_seconds=0.0))
    (r0, s0, cam0, f0), (r1, s1, cam1, f1) = results
    assert s0["termination"] == s1["termination"] == s_o["termination"]
    assert s0["iterations"] == s1["iterations"] == s_o["iterations"]
    assert np.isclose(s0["initial_cost"], s_o["initial_cost"], rtol=1e-12)
    assert abs(s0["final_cost"] - s_o["final_cost"]) <= 1e-9 * s_o["final_cost"]
    assert s0["final_cost"] == s1["final_cost"]                 # identical decisions on both ranks
    assert np.array_equal(cam0, cam1) and f0 == f1              # replicated cameras stay bit-identical
    assert np.allclose(cam0, cam_o, atol=1e-8) and np.isclose(f0, f_o, rtol=1e-10)


def test_shard_points_partition(sfm):
    prob = sfm.make_problem("small")
    parts = [prob.shard_points(r, 3) for r in range(3)]
    assert sum(p.n_obs for p in parts) == prob.n_obs and sum(p.n_pt for p in parts) == prob.n_pt
    for p in parts:
        assert p.n_cam == prob.n_cam and np.array_equal(p.cam6, prob.cam6)
        assert p.obs_pt.min() >= 0 and p.obs_pt.max() < p.n_pt
