import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from sfm_toy_library_amd import capi
rng = np.random.default_rng(0)
n = 1201
B = rng.standard_normal((n, n)); A = B @ B.T + n * np.eye(n); b = rng.standard_normal(n)
for _ in range(3): capi.dense_spd_solve(A, b, method=0)
t = time.perf_counter()
for _ in range(10): x, info, it = capi.dense_spd_solve(A, b, method=0)
print("dense_spd_solve %.2f ms per call (incl. H2D/D2H)" % (1e3 * (time.perf_counter() - t) / 10), "resid", np.abs(A @ x - b).max())

