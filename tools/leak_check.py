"""Create / solve / append / destroy in a loop: resident set size of the process and free device memory must level off (the chunk cache and the
host kits are bounded: device_arena.hip CACHE_LIMIT / KIT_LIMIT).      python tools/leak_check.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import psutil
import torch
import sfm_toy_library_amd as sfm
from sfm_toy_library_amd import capi

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
proc = psutil.Process()
rng = np.random.default_rng(5)
probs = [sfm.make_problem("cfg2", n_cam=int(c), n_pt=int(p), views=(2, 6), seed=int(s)) for c, p, s in [(5, 300, 1), (40, 4000, 2), (90, 9000, 3), (230, 6000, 4), (14, 900, 5)]]
big = sfm.make_problem("cfg3", n_cam=60, n_pt=30000, seed=9)
t0 = time.time()
first = None
for r in range(rounds):
    prob = probs[r % len(probs)] if r % 10 else big
    precision = r % 2
    flags = [0, sfm.CREATE_DETERMINISTIC, 0, sfm.CREATE_NO_PAIR_LIST][r % 4]
    with capi.Problem(prob, precision=precision, flags=flags) as P:
        P.solve(capi.default_options(max_seconds=0.0, precision=precision, linear_solver=r % 3 if not flags & sfm.CREATE_NO_PAIR_LIST else 2))
        P.reset()
        P.solve(capi.default_options(max_seconds=0.0, precision=precision))
    if r % 3 == 0:
        capi.solve(probs[(r // 3) % len(probs)], capi.default_options(max_seconds=0.0, precision=precision))
    if r % 25 == 0 or r == rounds - 1:
        free, total = torch.cuda.mem_get_info(0)
        rss = proc.memory_info().rss / 2**20
        if first is None and r >= 50:
            first = (rss, free)
        print("round %4d  rss %8.1f MiB  device free %8.1f MiB  (%.0f s)" % (r, rss, free / 2**20, time.time() - t0), flush=True)
free, total = torch.cuda.mem_get_info(0)
rss = proc.memory_info().rss / 2**20
if first:
    print("drift since round 50: rss %+.1f MiB, device free %+.1f MiB" % (rss - first[0], (free - first[1]) / 2**20))
