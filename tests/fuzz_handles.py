"""Randomised sweep of the HANDLE variants of the C ABI against the oracle (test infrastructure -- it calls the oracle; GPU box), the companion of
tests/fuzz_parity.py: for random shapes
  * a handle that GROWS (sfmba_problem_append: the reference's BA after every added view, SfM.cpp:464-466) in one to four random steps -- after every
    step the solve must be what the oracle gives for that step's problem;
  * a DETERMINISTIC handle (SFMBA_CREATE_DETERMINISTIC) solved twice: bit-identical, and on the oracle;
  * a handle WITHOUT a pair list (SFMBA_CREATE_NO_PAIR_LIST: the matrix-free solve) on the oracle;
  * sfmba_problem_set_params with the oracle's solution: the next solve ends at once where the oracle would.

    python tests/fuzz_handles.py [--cases N] [--seed S]

Exit code 1 on any mismatch: termination, LM iteration count on runs of <= 40 iterations, final cost beyond 1e-7 (fp64, exact solver) / 1e-5 relative (fp32 Jacobians, CG at 1e-8: their own bars are held by the parity tests)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LONG_RUN = 40


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=150)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--big", action="store_true", help="shapes above 213 cameras as well (the streaming CG on one triangle; the oracle takes seconds each)")
    args = ap.parse_args()
    import sfm_toy_library_amd as sfm
    from sfm_toy_library_amd import capi
    from oracle import oracle_py as oracle
    oracle.set_num_threads(1)
    rng = np.random.default_rng(args.seed)
    bad = 0
    counts = {"append": 0, "deterministic": 0, "matrix-free": 0, "set_params": 0}
    t0 = time.time()

    def sub_problem(prob, order):
        return sfm.BAProblem(prob.cam6, prob.pt3, prob.focal, prob.obs_cam[order], prob.obs_pt[order], prob.obs_xy[order])

    def compare(tag, case, desc, s, s_o, exact):
        nonlocal bad
        scale = abs(s_o["final_cost"]) + 1e-14 * abs(s_o["initial_cost"]) + 1e-300
        rel = abs(s["final_cost"] - s_o["final_cost"]) / scale
        floor = 1e-12 * abs(s_o["initial_cost"]) + 1e-13          # (a cost of 1e-15 is the rounding of float32-rounded observations of an exactly satisfiable problem)
        tiny = s_o["final_cost"] <= 1e-9 * s_o["initial_cost"]            # exactly satisfiable: one iteration more or less at a cost of ~0
        long_run = s_o["iterations"] > LONG_RUN
        off = s["termination_name"] != s_o["termination_name"]
        if not tiny and not long_run:
            off = off or (exact and s["iterations"] != s_o["iterations"]) or (rel > (1e-7 if exact else 1e-5) and abs(s["final_cost"] - s_o["final_cost"]) > floor)
        if off:
            bad += 1
            print("case %d %s MISMATCH: %s | oracle %s it %d cost %.12e | hip %s it %d cost %.12e (rel %.2e)" % (
                case, tag, desc, s_o["termination_name"], s_o["iterations"], s_o["final_cost"], s["termination_name"], s["iterations"], s["final_cost"], rel))
        return not off

    for case in range(args.cases):
        n_cam = int(rng.choice([2, 3, 5, 8, 14, 25, 43, 44, 60, 90]))
        views = (2, min(n_cam, int(rng.integers(2, 9))))
        n_pt = int(rng.choice([40, 150, 600, 1500]))
        if args.big and rng.random() < 0.6:
            n_cam, n_pt = int(rng.choice([214, 230, 260])), int(rng.choice([2500, 5000]))
            views = (3, int(rng.integers(4, 9)))
        prob = sfm.make_problem("cfg2", n_cam=n_cam, n_pt=n_pt, views=views, seed=int(rng.integers(1, 1 << 30)), noise_px=float(rng.choice([0.0, 0.5, 2.0]))).copy()
        precision = int(rng.integers(0, 2))
        linear = int(rng.integers(0, 3))
        exact = precision == 0 and linear in (0, 2)
        opt = capi.default_options(max_seconds=0.0, precision=precision, linear_solver=linear)
        opt_o = sfm.SfmbaOptions.defaults(max_seconds=0.0)
        desc = "n_cam %d n_pt %d n_obs %d views %r precision %d linear %d" % (n_cam, n_pt, prob.n_obs, views, precision, linear)
        what = rng.choice(["append", "deterministic", "matrix-free", "set_params"])
        counts[str(what)] += 1
        try:
            if what == "append" and n_cam >= 3:
                # cameras registered in index order; a step's observations: every point seen by at least two registered cameras
                cuts = sorted(set(int(c) for c in rng.integers(2, n_cam, size=int(rng.integers(1, 4))))) + [n_cam]
                have = np.zeros(prob.n_obs, bool)
                P = None
                order = None
                for c in cuts:
                    vis = prob.obs_cam < c
                    cnt = np.bincount(prob.obs_pt[vis], minlength=prob.n_pt)
                    now = vis & (cnt[prob.obs_pt] >= 2)
                    new = np.flatnonzero(now & ~have)
                    have = now
                    if P is None:
                        if len(new) == 0:
                            continue
                        order = new
                        P = capi.Problem(sub_problem(prob, order), precision=precision)
                    else:
                        if len(new) == 0:
                            continue
                        P.append(prob.cam6, prob.pt3, prob.focal, prob.obs_cam[new], prob.obs_pt[new], prob.obs_xy[new])
                        order = np.concatenate([order, new])
                    s, _ = P.solve(opt)
                    s_o = oracle.solve(sub_problem(prob, order), opt_o)[3]
                    compare("append (cameras < %d)" % c, case, desc, s, s_o, exact)
                if P is not None:
                    P.close()
            elif what == "deterministic":
                s_o = oracle.solve(prob, opt_o)[3]
                with capi.Problem(prob, precision=precision, flags=sfm.CREATE_DETERMINISTIC) as P:
                    s1, _ = P.solve(opt)
                    c1, p1, f1 = P.get_params()
                    P.reset()
                    s2, _ = P.solve(opt)
                    c2, p2, f2 = P.get_params()
                if s1["final_cost"] != s2["final_cost"] or s1["iterations"] != s2["iterations"] or not np.array_equal(c1, c2) or not np.array_equal(p1, p2) or f1 != f2:
                    bad += 1
                    print("case %d deterministic handle NOT bit-identical: %s | %r it %d vs %r it %d" % (case, desc, s1["final_cost"], s1["iterations"], s2["final_cost"], s2["iterations"]))
                compare("deterministic", case, desc, s1, s_o, exact)
            elif what == "matrix-free":
                s_o = oracle.solve(prob, opt_o)[3]
                with capi.Problem(prob, precision=precision, flags=sfm.CREATE_NO_PAIR_LIST) as P:
                    s, _ = P.solve(capi.default_options(max_seconds=0.0, precision=precision))
                compare("matrix-free", case, desc, s, s_o, False)
            else:
                cam_o, pt_o, f_o, s_o, _ = oracle.solve(prob, opt_o)
                with capi.Problem(prob, precision=precision) as P:
                    P.solve(opt)
                    P.set_params(cam_o, pt_o, f_o)
                    s, _ = P.solve(opt)
                at = sub_problem(prob, np.arange(prob.n_obs))
                at.cam6, at.pt3, at.focal = cam_o.copy(), pt_o.copy(), float(f_o)
                s_at = oracle.solve(at, opt_o)[3]
                compare("set_params", case, desc, s, s_at, exact)
        except Exception as e:
            bad += 1
            print("case %d %s EXCEPTION %s: %s | %s" % (case, what, type(e).__name__, e, desc))
    print("fuzz_handles: %d cases (%s): %d mismatches, %.0f s" % (args.cases, ", ".join("%s %d" % kv for kv in sorted(counts.items())), bad, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
