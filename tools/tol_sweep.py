"""CG tolerance of the two-level PCG against time to solution and against the exact (Cholesky) solve: LM iterations, CG iterations,
ms per resident solve, final RMS and the parameter-level difference.  usage: tol_sweep.py [workload ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sfm_toy_library_amd as sfm
from sfm_toy_library_amd import capi

for name in sys.argv[1:] or ["cfg3", "cfg3_banded", "cfg2"]:
    prob = sfm.make_problem(name)
    prec = 1 if prob.n_obs > 100000 else 0
    P = capi.Problem(prob, precision=prec)
    ref_s, _ = P.solve(capi.default_options(max_seconds=0.0, linear_solver=0, precision=prec))
    rc, rp, rf = P.get_params()
    rms_ref = float(np.sqrt(2.0 * ref_s["final_cost"] / prob.n_obs))
    print("%-12s exact: LM %d  cost %.10e  rms %.9f px" % (name, ref_s["iterations"], ref_s["final_cost"], rms_ref), flush=True)
    for anchored in (1, 0):
        for tol in (1e-8, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1):
            o = capi.default_options(max_seconds=0.0, linear_solver=1, pcg_tolerance=tol, pcg_anchored=anchored, precision=prec)
            for _ in range(3):
                P.reset(); P.solve(o)
            torch.cuda.synchronize()
            n = 20
            t0 = time.perf_counter()
            for _ in range(n):
                P.reset(); s, _ = P.solve(o)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / n
            c, p, f = P.get_params()
            rms = float(np.sqrt(2.0 * s["final_cost"] / prob.n_obs))
            print("%-12s anchored %d tol %.0e: LM %2d  cg %4d  %.3f ms/solve  %7.1f it/s  d_rms %.2e px  |dcam| %.1e |dpt| %.1e |df| %.1e  %s" % (
                name, anchored, tol, s["iterations"], s["linear_iters"], ms, 1e3 * s["iterations"] / ms, abs(rms - rms_ref),
                np.abs(c - rc).max(), np.abs(p - rp).max(), abs(f - rf), s["termination_name"]), flush=True)
    P.close()
