import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
import sfm_toy_library_amd as sfm
from sfm_toy_library_amd import capi
prob = sfm.make_problem("cfg2", n_cam=1, n_pt=30, views=(1, 1), seed=11)
for prec, lin in ((0,1),(1,1),(1,2),(1,0)):
    kw = dict(max_seconds=0.0, max_iters=30, precision=prec, linear_solver=lin, verbose=1)
    if lin == 1: kw.update(pcg_tolerance=1e-12, pcg_anchored=0)
    cam, pt, f, s, tr = capi.solve(prob, capi.default_options(**kw))
    print(prec, lin, s["termination_name"], s["message"], s["iterations"], s["final_cost"])
