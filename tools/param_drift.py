"""Parameter-level difference between the exact (Cholesky) reduced solve and PCG at several tolerances."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sfm_toy_library_amd as sfm
from sfm_toy_library_amd import capi
for name in sys.argv[1:] or ["tiny", "crazyhorse_like", "cfg2", "cfg3"]:
    prob = sfm.make_problem(name)
    ref = capi.solve(prob, capi.default_options(max_seconds=0.0, linear_solver=0))
    for tol in (1e-6, 1e-7, 1e-8, 1e-10):
        r = capi.solve(prob, capi.default_options(max_seconds=0.0, linear_solver=1, pcg_tolerance=tol))
        print("%-16s tol %.0e: iters %d/%d  lin %3d  |dcam| %.2e  |dpt| %.2e  |df| %.2e  dcost/cost %.1e" % (
            name, tol, r[3]["iterations"], ref[3]["iterations"], r[3]["linear_iters"], np.abs(r[0] - ref[0]).max(), np.abs(r[1] - ref[1]).max(),
            abs(r[2] - ref[2]), abs(r[3]["final_cost"] - ref[3]["final_cost"]) / ref[3]["final_cost"]), flush=True)
