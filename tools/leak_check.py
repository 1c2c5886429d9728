"""Device-memory stability over many one-shot calls with varying sizes (the incremental caller's pattern)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sfm_toy_library_amd as sfm
from sfm_toy_library_amd import capi
hip = C.CDLL("libamdhip64.so")
def free_mb():
    f, t = C.c_size_t(0), C.c_size_t(0)
    hip.hipMemGetInfo(C.byref(f), C.byref(t))
    return f.value / 2**20
capi.solve(sfm.make_problem("tiny"))
base = free_mb()
lo = base
for k in range(120):
    n_pt = 500 + 97 * k                       # a growing reconstruction
    prob = sfm.make_problem("cfg2", n_cam=8 + k // 10, n_pt=n_pt, seed=k)
    s = capi.solve(prob, capi.default_options(max_seconds=0.0, precision=1, linear_solver=2))[3]
    assert s["termination_name"] == "CONVERGENCE", s
    lo = min(lo, free_mb())
print("free at start %.0f MB, lowest during 120 growing one-shot solves %.0f MB (cache holds %.0f MB)" % (base, lo, base - free_mb()))
released = capi.release_cache() / 2**20
print("release_cache returned %.0f MB; free now %.0f MB (start %.0f MB)" % (released, free_mb(), base))
