"""Pair-pass time for the three lane-group sizes on a problem shape given as n_cam n_pt (views 10)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sfm_toy_library_amd as sfm
from sfm_toy_library_amd import capi
n_cam, n_pt = int(sys.argv[1]), int(sys.argv[2])
prob = sfm.make_problem("cfg5", n_cam=n_cam, n_pt=n_pt)
for lpb in ("64", "16"):
    os.environ["SFMBA_PAIR_LPB"] = lpb
    with capi.Problem(prob, precision=1) as P:
        opt = capi.default_options(max_seconds=0.0, precision=1, linear_solver=1)
        P.solve(opt); P.set_profiling(True); P.reset(); s, _ = P.solve(opt); prof = P.get_profile()
        print("n_cam %d n_pt %d (mean pairs/block %.1f) lpb %s: pairs %.1f us, solve cost %.8e %s" % (
            n_cam, n_pt, prob.n_obs * 4.5 / (n_cam * (n_cam - 1) / 2), lpb, prof["schur_pairs"]["avg_us"], s["final_cost"], s["termination_name"]), flush=True)
