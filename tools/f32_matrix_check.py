"""Streaming CG path (d > 1280) with the preconditioned matrix stored in fp32 vs fp64: cost, parameters, time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sfm_toy_library_amd as sfm
from sfm_toy_library_amd import capi
name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
kw = dict(n_cam=int(sys.argv[2]), n_pt=int(sys.argv[3])) if len(sys.argv) > 3 else {}
prob = sfm.make_problem(name, **kw)
res = {}
for mode in ("0", "1"):
    os.environ["SFMBA_PCG_F32_MATRIX"] = mode
    with capi.Problem(prob, precision=1) as P:
        opt = capi.default_options(max_seconds=0.0, precision=1, linear_solver=1)
        P.solve(opt); P.reset()
        t0 = time.perf_counter(); s, tr = P.solve(opt); dt = time.perf_counter() - t0
        res[mode] = (s, P.get_params(), dt)
        print("fp32 matrix" if mode == "1" else "fp64 matrix", "%.3f ms" % (1e3 * dt), s["termination_name"], s["iterations"], "lin", s["linear_iters"], "cost %.10e" % s["final_cost"], flush=True)
a, b = res["0"], res["1"]
print("rel cost diff %.2e  |dcam| %.2e  |dpt| %.2e" % (abs(a[0]["final_cost"] - b[0]["final_cost"]) / a[0]["final_cost"], np.abs(a[1][0] - b[1][0]).max(), np.abs(a[1][1] - b[1][1]).max()))
