"""Diagnostic: the (world, case) of tests/test_gpu_sharded.py::test_row_sharded_hip_solve with a watchdog that dumps every rank's Python stack.
    python tools/debug_row_hang.py 4 row_wide [seconds]"""
import faulthandler, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch.multiprocessing as mp


def worker(rank, world, port, case, out, secs):
    faulthandler.dump_traceback_later(secs, exit=True, file=sys.stderr)
    import test_gpu_sharded as t
    t.CASES[case][3]["verbose"] = 1
    t._worker(rank, world, port, case, out, True, int(os.environ.get('DBG_REPEATS', '2')), 0)


if __name__ == "__main__":
    world, case = int(sys.argv[1]), sys.argv[2]
    secs = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, 29655, case, out, secs)) for r in range(world)]
    for p in procs: p.start()
    try:
        res = [out.get(timeout=secs + 30) for _ in range(world)]
        print("OK", [(r[0], r[1]["iterations"], r[1]["final_cost"], r[1]["linear_iters"]) for r in res])
    except Exception as e:
        print("FAILED", type(e).__name__)
    for p in procs:
        p.join(timeout=10)
        if p.is_alive(): p.kill()
