"""Functional + timing check of BASELINE config 5 size (1000 cams / 500k pts / 5M obs) on one GPU."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sfm_toy_library_amd as sfm
from sfm_toy_library_amd import capi
t0 = time.time(); prob = sfm.make_problem("cfg5"); print("generate %.1fs" % (time.time() - t0), prob.n_cam, prob.n_pt, prob.n_obs, flush=True)
t0 = time.time(); P = capi.Problem(prob, precision=1); print("create (structure build + H2D) %.1fs" % (time.time() - t0), "d =", P.reduced_dim, flush=True)
for lin in (1,):
    opt = capi.default_options(max_seconds=0.0, precision=1, linear_solver=lin)
    for r in range(2):
        P.reset(); t0 = time.time(); s, tr = P.solve(opt); dt = time.time() - t0
        print("linear", lin, "solve %.3fs" % dt, s["termination_name"], s["iterations"], "lin_iters", s["linear_iters"], "cost %.6e" % s["final_cost"],
              "rms %.6f" % np.sqrt(2 * s["final_cost"] / prob.n_obs), [round(r_["cost"], 1) for r_ in tr], flush=True)
P.set_profiling(True); P.reset(); P.solve(opt); prof = P.get_profile(); P.set_profiling(False)
print({k: round(v["avg_us"], 1) for k, v in prof.items()})
