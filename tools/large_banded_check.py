"""Where the d <= 1280 limit of the segmented coarse space leaves a LONG camera path: 600 cameras / 300k points / ~3M observations, banded
(d = 3601: the streaming CG kernels, eight global vectors).     python tools/large_banded_check.py [n_cam] [n_pt]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sfm_toy_library_amd as sfm
from sfm_toy_library_amd import capi

n_cam = int(sys.argv[1]) if len(sys.argv) > 1 else 600
n_pt = int(sys.argv[2]) if len(sys.argv) > 2 else 300000
prob = sfm.make_problem("cfg3_banded", n_cam=n_cam, n_pt=n_pt)
print("banded: %d cams / %d pts / %d obs, d = %d" % (prob.n_cam, prob.n_pt, prob.n_obs, 6 * prob.n_cam + 1))
with capi.Problem(prob, precision=1) as P:
    for name, opt in (("auto", capi.default_options(max_seconds=0.0, precision=1)),
                      ("pcg1e-8", capi.default_options(max_seconds=0.0, precision=1, linear_solver=1)),
                      ("pcg1e-3", capi.default_options(max_seconds=0.0, precision=1, linear_solver=1, pcg_tolerance=1e-3)),
                      ("cholesky", capi.default_options(max_seconds=0.0, precision=1, linear_solver=0))):
        P.reset(); s, tr = P.solve(opt)
        t = []
        for _ in range(3):
            P.reset(); t0 = time.perf_counter(); s, tr = P.solve(opt); t.append(time.perf_counter() - t0)
        print("%-9s %d LM its, CG %s, cost %.10e, %.2f ms -> %.0f LM it/s" % (name, s["iterations"], [r["linear_iters"] for r in tr[1:]], s["final_cost"], 1e3 * min(t), s["iterations"] / min(t)))
