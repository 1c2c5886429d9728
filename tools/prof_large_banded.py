import os, sys
sys.path.insert(0, "/root/repo")
import sfm_toy_library_amd as sfm
from sfm_toy_library_amd import capi
prob = sfm.make_problem("cfg3_banded", n_cam=600, n_pt=300000)
with capi.Problem(prob, precision=1) as P:
    opt = capi.default_options(max_seconds=0.0, precision=1, linear_solver=1)
    for _ in range(3):
        P.reset(); P.solve(opt)
