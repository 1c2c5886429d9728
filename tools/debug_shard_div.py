"""Two ranks on one GPU, the native sharded loop on the 'wide' case, repeated: do the replicated cameras stay bit-identical?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch.multiprocessing as mp


def worker(rank, world, port, trials, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sfm_toy_library_amd as sfm
    from sfm_toy_library_amd import capi
    from sfm_toy_library_amd.sharded import HipShardBackend, solve_sharded_native
    prob = sfm.make_problem(name="cfg3", n_cam=230, n_pt=6000, seed=77)
    be = HipShardBackend(prob, rank, world, device=0, precision=1)
    opt = capi.default_options(max_seconds=0.0, precision=1, linear_solver=1)
    res = []
    for t in range(trials):
        be.reset()
        s = solve_sharded_native(be, opt, dist=dist)
        cam, pt, f = be.get_params()
        res.append((s["iterations"], s["linear_iters"], s["final_cost"], cam.copy(), f))
    out.put((rank, res))
    dist.barrier(); be.close(); dist.destroy_process_group()


if __name__ == "__main__":
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    ctx = mp.get_context("spawn"); out = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, 29900 + os.getpid() % 90, trials, out)) for r in range(2)]
    for p in procs: p.start()
    r = dict(out.get(timeout=240) for _ in range(2))
    for p in procs: p.join(timeout=60)
    bad = 0
    for t in range(trials):
        a, b = r[0][t], r[1][t]
        same = np.array_equal(a[3], b[3]) and a[4] == b[4]
        ref = r[0][0]
        drift = np.abs(a[3] - ref[3]).max()
        if not same or a[:3] != b[:3]:
            bad += 1
            d = np.abs(a[3] - b[3])
            print("trial %d: ranks differ: iters %s/%s lin %s/%s cost %.12e/%.12e max|dcam| %.3e at %s; vs trial 0: %.3e" % (
                t, a[0], b[0], a[1], b[1], a[2], b[2], d.max(), np.unravel_index(d.argmax(), d.shape), drift))
        elif drift > 0:
            print("trial %d: ranks agree but differ from trial 0 by %.3e (lin %s vs %s)" % (t, drift, a[1], ref[1]))
    print("trials %d, rank disagreements %d" % (trials, bad))
