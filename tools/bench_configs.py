"""Resident-solve timing of the other BASELINE configurations (parity-test cases, not bench lines)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sfm_toy_library_amd as sfm
from sfm_toy_library_amd import capi
rows = [("crazyhorse_like", 0), ("cfg2", 0), ("cfg2", 1), ("cfg4", 1), ("cfg3", 0), ("cfg3", 1)]
for name, prec in rows:
    prob = sfm.make_problem(name, sub=0 if name == "cfg4" else None)
    for lin in (1, 0):
        with capi.Problem(prob, precision=prec) as P:
            opt = capi.default_options(max_seconds=0.0, precision=prec, linear_solver=lin)
            for _ in range(3):
                P.reset(); P.solve(opt)
            n, t0, its = 20, time.perf_counter(), 0
            for _ in range(n):
                P.reset(); s, _ = P.solve(opt); its += s["iterations"]
            dt = (time.perf_counter() - t0) / n
            print("%-16s %-4s %-8s cams %4d pts %6d obs %7d : %.3f ms/solve, %d LM iterations, %.0f it/s, RMS %.6f px, %s"
                  % (name, "f32j" if prec else "f64", "pcg" if lin else "cholesky", prob.n_cam, prob.n_pt, prob.n_obs, 1e3 * dt, s["iterations"],
                     its / n / dt, np.sqrt(2 * s["final_cost"] / prob.n_obs), s["termination_name"]), flush=True)
