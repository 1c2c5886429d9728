"""AUTO's CG tolerance (SFMBA_AUTO_TOL, experiment) against the exact (Cholesky) solve: parameters, cost, time.  One process per tolerance."""
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import time, numpy as np, torch
    import sfm_toy_library_amd as sfm
    from sfm_toy_library_amd import capi
    for name, prec in (("cfg3", 1), ("cfg3", 0), ("cfg3_banded", 1), ("cfg2", 0)):
        prob = sfm.make_problem(name)
        with capi.Problem(prob, precision=prec) as P:
            P.solve(capi.default_options(max_seconds=0.0, linear_solver=0, precision=prec)); rc, rp, rf = P.get_params()
            o = capi.default_options(max_seconds=0.0, precision=prec)
            for _ in range(3): P.reset(); s, _ = P.solve(o)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): P.reset(); s, _ = P.solve(o)
            torch.cuda.synchronize(); ms = 1e2 * (time.perf_counter() - t0)
            c, p, f = P.get_params()
            print("tol %s %-12s prec %d: LM %d cg %3d fallbacks %d  %.3f ms  |dcam| %.1e |dpt| %.1e |df| %.1e" % (os.environ.get("SFMBA_AUTO_TOL"), name, prec, s["iterations"], s["linear_iters"], s["cholesky_fallbacks"], ms, np.abs(c - rc).max(), np.abs(p - rp).max(), abs(f - rf)), flush=True)
else:
    for tol in ("1e-12", "1e-11", "1e-10", "1e-9"):
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, SFMBA_AUTO_TOL=tol))
