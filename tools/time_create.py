"""Host-side cost of the one-shot boundary: structure build + H2D vs the resident solve (cfg 3 by default)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sfm_toy_library_amd as sfm
from sfm_toy_library_amd import capi
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
prob = sfm.make_problem(name)
for r in range(3):
    t0 = time.perf_counter(); P = capi.Problem(prob, precision=1); t1 = time.perf_counter()
    opt = capi.default_options(max_seconds=0.0, precision=1, linear_solver=1)
    s, _ = P.solve(opt); t2 = time.perf_counter()
    P.reset(); s, _ = P.solve(opt); t3 = time.perf_counter()
    P.close(); t4 = time.perf_counter()
    print("create %.1f ms  first solve %.2f ms  resident solve %.2f ms  destroy %.1f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3)), flush=True)
for r in range(3):
    t0 = time.perf_counter()
    cam, pt, f, s, tr = capi.solve(prob, capi.default_options(max_seconds=0.0, precision=1, linear_solver=1))
    print("one-shot sfmba_solve %.1f ms  (%s, %d iterations)" % (1e3 * (time.perf_counter() - t0), s["termination_name"], s["iterations"]), flush=True)
