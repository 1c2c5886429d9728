"""The incremental loop on a long camera path: 599 views resident, the 600th appended (sfmba_problem_append) -- the rebuilt structure chooses the pair-pass
geometry from the fill of the previous one.    python tools/large_banded_append_check.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sfm_toy_library_amd as sfm
from sfm_toy_library_amd import capi

prob = sfm.make_problem("cfg3_banded", n_cam=600, n_pt=300000)
last = prob.n_cam - 1
old = np.nonzero(prob.obs_cam != last)[0]
new = np.nonzero(prob.obs_cam == last)[0]
first = sfm.BAProblem(prob.cam6[:last], prob.pt3, prob.focal, prob.obs_cam[old], prob.obs_pt[old], prob.obs_xy[old])
opt = capi.default_options(max_seconds=0.0, precision=1, linear_solver=1)
with capi.Problem(first, precision=1) as P:
    s, _ = P.solve(opt)
    print("599 views: %d LM its, cost %.8e" % (s["iterations"], s["final_cost"]))
    P.reset()
    t0 = time.perf_counter()
    P.append(prob.cam6, prob.pt3, prob.focal, prob.obs_cam[new], prob.obs_pt[new], prob.obs_xy[new])
    t1 = time.perf_counter()
    s, tr = P.solve(opt)
    t = []
    for _ in range(3):
        P.reset(); t2 = time.perf_counter(); s, tr = P.solve(opt); t.append(time.perf_counter() - t2)
    print("600 views after the append (%.1f ms): %d LM its, CG %s, cost %.10e, %.2f ms -> %.0f LM it/s" % (1e3 * (t1 - t0), s["iterations"], [r["linear_iters"] for r in tr[1:]], s["final_cost"], 1e3 * min(t), s["iterations"] / min(t)))
with capi.Problem(prob, precision=1) as P:
    s, tr = P.solve(opt)
    t = []
    for _ in range(3):
        P.reset(); t2 = time.perf_counter(); s, tr = P.solve(opt); t.append(time.perf_counter() - t2)
    print("600 views built in one go: %d LM its, cost %.10e, %.2f ms -> %.0f LM it/s" % (s["iterations"], s["final_cost"], 1e3 * min(t), s["iterations"] / min(t)))
