"""Stress of the dense exact solver through the C ABI: sizes around every tile / path boundary, repeated, against numpy."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from sfm_toy_library_amd import capi
rng = np.random.default_rng(0)
sizes = list(range(1, 6)) + list(range(60, 70)) + list(range(124, 132)) + list(range(188, 196)) + [255, 256, 257, 511, 512, 640, 1023, 1024, 1279, 1280, 1281] + \
        list(range(2555, 2565)) + [3000]
worst = 0.0
for rep in range(2):
    for n in sizes:
        M = rng.normal(size=(n, n)); A = M @ M.T + n * np.eye(n); b = rng.normal(size=n)
        x, info, _ = capi.dense_spd_solve(A, b, method=0)
        ref = np.linalg.solve(A, b)
        err = np.abs(x - ref).max() / np.abs(ref).max()
        worst = max(worst, err)
        assert info == 0 and err < 1e-9, (n, info, err)
print("sizes", len(sizes), "x2 ok, worst relative error %.2e" % worst)
