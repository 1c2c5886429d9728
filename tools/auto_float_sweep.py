"""what would SFMBA_LINEAR_AUTO lose if its CG stopped before 1e-12?  adjustBundle() narrows everything it writes back to float
(BA.cpp:201-221: pose = float(R(w)), float(t); points float; focal float).  For each fixture: the always-factorised solve as the reference, then the
two-level CG (segmented where the structure says so) at a plain relative tolerance t -- the arithmetic of AUTO's CG at that tolerance -- and the
Hamming distance between the written-back FLOAT containers (values that differ at all, and by more than one ulp), with the time per solve.
    python tools/auto_float_sweep.py [workload ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sfm_toy_library_amd as sfm
from sfm_toy_library_amd import capi
from sfm_toy_library_amd.synthetic import rotvec_to_matrix


def containers(cam, pt, f):
    R = rotvec_to_matrix(cam[:, :3]).astype(np.float32).reshape(len(cam), 9)
    return np.concatenate([R.ravel(), cam[:, 3:].astype(np.float32).ravel(), pt.astype(np.float32).ravel(), np.float32([f])])


def ulps(a, b):
    ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, np.int64(-2**31) - ia, ia); ib = np.where(ib < 0, np.int64(-2**31) - ib, ib)
    return np.abs(ia - ib)


names = sys.argv[1:] or ["crazyhorse_like", "cfg2", "cfg3", "cfg3_banded"]
for name in names:
    prob = sfm.make_problem(name)
    for prec in (0, 1):
        with capi.Problem(prob, precision=prec) as P:
            def run(opt, reps=10):
                for _ in range(2):
                    P.reset(); P.solve(opt)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(reps):
                    P.reset(); s, _ = P.solve(opt)
                torch.cuda.synchronize()
                return s, 1e3 * (time.perf_counter() - t0) / reps, P.get_params()
            s0, ms0, (c0, p0, f0) = run(capi.default_options(max_seconds=0.0, linear_solver=0, precision=prec))
            ref = containers(c0, p0, f0)
            sa, msa, (ca, pa, fa) = run(capi.default_options(max_seconds=0.0, linear_solver=2, precision=prec))
            u = ulps(containers(ca, pa, fa), ref)
            print("%-16s %s  CHOLESKY %.3f ms (%d LM its) | AUTO (today) %.3f ms, cg %d, fallbacks %d, floats differing %d of %d (>1 ulp: %d)" %
                  (name, "f64 " if prec == 0 else "f32j", ms0, s0["iterations"], msa, sa["linear_iters"], sa["cholesky_fallbacks"], int((u > 0).sum()), u.size, int((u > 1).sum())), flush=True)
            for tol in (1e-12, 1e-11, 1e-10, 1e-9, 1e-8, 1e-6):
                s, ms, (c, p, f) = run(capi.default_options(max_seconds=0.0, linear_solver=1, precision=prec, pcg_tolerance=tol, pcg_anchored=0))
                u = ulps(containers(c, p, f), ref)
                print("    CG to %.0e: %.3f ms  LM %d  cg %3d  floats differing %6d (>1 ulp: %d, max %d ulp)  |dcost| %.1e" %
                      (tol, ms, s["iterations"], s["linear_iters"], int((u > 0).sum()), int((u > 1).sum()), int(u.max()), abs(s["final_cost"] - s0["final_cost"]) / s0["final_cost"]), flush=True)
