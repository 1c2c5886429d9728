"""Camera counts the suite had never seen (VERDICT r5 "weak" 10: nothing tested above 1 000 cameras while d * ld crosses 2^31 at ~7 700): banded problems of
n_cam cameras (cameras on a closed path, tracks of 2 .. 30 neighbouring views) solved in F32J + PCG, fp64 + PCG and, where it is affordable, with the
factorisation; the three must agree.      python tools/large_cameras_check.py n_cam [n_pt] [modes: f32j_pcg,f64_pcg,f64_chol] [max LM iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sfm_toy_library_amd as sfm
from sfm_toy_library_amd import capi

n_cam = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
n_pt = int(sys.argv[2]) if len(sys.argv) > 2 else 6 * n_cam
modes = (sys.argv[3] if len(sys.argv) > 3 else "f32j_pcg,f64_pcg,f64_chol").split(",")
max_iters = int(sys.argv[4]) if len(sys.argv) > 4 else 30
t0 = time.perf_counter()
prob = sfm.make_problem("cfg3_banded", n_cam=n_cam, n_pt=n_pt, seed=4242)
d = 6 * prob.n_cam + 1
ld = (d + 1 + 63) // 64 * 64
print("banded: %d cams / %d pts / %d obs, d = %d, d * ld = %.3e (2^31 = 2.147e9), generated in %.1f s" % (prob.n_cam, prob.n_pt, prob.n_obs, d, float(d) * ld, time.perf_counter() - t0), flush=True)
res = {}
for mode in modes:
    prec = 1 if mode.startswith("f32j") else 0
    lin = 0 if mode.endswith("chol") else 1
    t0 = time.perf_counter()
    try:
        with capi.Problem(prob, precision=prec) as P:
            t1 = time.perf_counter()
            s, tr = P.solve(capi.default_options(max_seconds=0.0, precision=prec, linear_solver=lin, max_iters=max_iters))
            t2 = time.perf_counter()
            cam, pt, f = P.get_params()
        res[mode] = (cam, pt, f, s)
        print("%-9s create %.2f s, solve %.2f s: %s, %d LM its, CG %s, cost %.10e -> %.10e, focal %.6f" % (
            mode, t1 - t0, t2 - t1, s["termination_name"], s["iterations"], [r["linear_iters"] for r in tr[1:]], s["initial_cost"], s["final_cost"], f), flush=True)
    except Exception as e:
        print("%-9s FAILED: %s" % (mode, e), flush=True)
    capi.release_cache()
keys = list(res)
for a in range(len(keys)):
    for b in range(a + 1, len(keys)):
        ra, rb = res[keys[a]], res[keys[b]]
        print("%s vs %s: cost rel %.2e, cameras %.2e, points (99.9 %%) %.2e, focal rel %.2e, LM its %d / %d" % (
            keys[a], keys[b], abs(ra[3]["final_cost"] - rb[3]["final_cost"]) / rb[3]["final_cost"], np.abs(ra[0] - rb[0]).max(),
            np.quantile(np.linalg.norm(ra[1] - rb[1], axis=1), 0.999), abs(ra[2] - rb[2]) / rb[2], ra[3]["iterations"], rb[3]["iterations"]))
