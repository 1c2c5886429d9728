"""Throughput of sfmba_triangulate (host arrays in, host arrays out) vs the numpy oracle, n matches.  Lives under tests/ because it
runs the oracle (test infrastructure); not collected by pytest."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sfm_toy_library_amd as sfm
from sfm_toy_library_amd import capi
from oracle import triangulate_oracle as tri
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_gpu_triangulate import _random_scene
for n in (5000, 1000000):
    K, Pl, Pr, l, r, _ = _random_scene(n, 3)
    capi.triangulate(K, Pl, Pr, l, r)
    t0 = time.perf_counter(); reps = 20
    for _ in range(reps): X, keep, err = capi.triangulate(K, Pl, Pr, l, r)
    dt = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter(); Xo, ko, _, _ = tri.triangulate_views(K, Pl, Pr, l, r); dto = time.perf_counter() - t0
    print("n = %d: GPU call %.3f ms (%.1f M matches/s, host arrays both ways), numpy oracle %.1f ms; kept %d / %d" % (n, 1e3 * dt, n / dt / 1e6, 1e3 * dto, keep.sum(), n))
