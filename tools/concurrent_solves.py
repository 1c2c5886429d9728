"""Throughput of several independent problems solved concurrently on ONE GPU (one host thread + one resident problem
each; every problem has its own stream).  BASELINE config 4 = the cfg-3 problem as 8 independent sub-problems."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sfm_toy_library_amd as sfm
from sfm_toy_library_amd import capi
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
nprob = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = 30
probs = [sfm.make_problem(name, sub=g) for g in range(nprob)]
P = [capi.Problem(p, precision=1) for p in probs]
opt = capi.default_options(max_seconds=0.0, precision=1, linear_solver=1)
for h in P:
    h.solve(opt)
def work(h, out, k):
    its = 0
    for _ in range(reps):
        h.reset(); s, _ = h.solve(opt); its += s["iterations"]
    out[k] = its
# sequential
t0 = time.perf_counter(); out = [0] * nprob
for k, h in enumerate(P): work(h, out, k)
dt_seq = time.perf_counter() - t0
# concurrent
t0 = time.perf_counter(); out2 = [0] * nprob
th = [threading.Thread(target=work, args=(h, out2, k)) for k, h in enumerate(P)]
[t.start() for t in th]; [t.join() for t in th]
dt_con = time.perf_counter() - t0
print("%s x %d: sequential %.1f LM it/s (%.3f ms/solve), concurrent %.1f LM it/s (%.3f ms/solve/problem)" % (
    name, nprob, sum(out) / dt_seq, 1e3 * dt_seq / (reps * nprob), sum(out2) / dt_con, 1e3 * dt_con / reps))
