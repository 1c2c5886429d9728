import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import sfm_toy_library_amd as sfm
from sfm_toy_library_amd import capi
prob = sfm.make_problem("cfg3_banded")
for seg in ("0", "1"):
    os.environ["SFMBA_PCG_SEGMENTS"] = seg
    for prec in (1, 0):
        with capi.Problem(prob, precision=prec) as P:
            for name, opt in (("auto", capi.default_options(max_seconds=0.0, precision=prec)), ("pcg1e-8", capi.default_options(max_seconds=0.0, precision=prec, linear_solver=1)),
                              ("pcg1e-3", capi.default_options(max_seconds=0.0, precision=prec, linear_solver=1, pcg_tolerance=1e-3))):
                P.reset(); s, tr = P.solve(opt)
                t = []
                for _ in range(8):
                    P.reset(); t0 = time.perf_counter(); s, tr = P.solve(opt); t.append(time.perf_counter() - t0)
                print("segments %s prec %d %-8s: %d LM its, CG %s, fallbacks %d, cost %.10e, %.3f ms -> %.0f it/s" % (seg, prec, name, s["iterations"], [r["linear_iters"] for r in tr[1:]], s["cholesky_fallbacks"], s["final_cost"], 1e3 * min(t), s["iterations"] / min(t)))
