"""Randomised parity sweep of the HIP path against the oracle (test infrastructure -- it calls the oracle, so it lives under tests/; GPU box): random shapes around every dispatch threshold of the
library (Cholesky / CG switch of AUTO at 256 reduced unknowns, the register-resident CG up to d = 1280, the streaming symmetric CG above, the
banded / segmented forms), track lengths 1 .. 12, gross outliers (rejected LM steps), large initial perturbations, both precisions, all three
linear-solver settings, the resident handle solved twice (reset) and the one-shot entry point.

    python tests/fuzz_parity.py [--cases N] [--seed S] [--big]

Prints one line per mismatch and a summary.  HARD (exit code 1): the exact path -- fp64 with the factorisation or AUTO, what a drop-in caller of
adjustBundle() runs -- ends elsewhere than the oracle (termination, LM iteration count, or final cost beyond 1e-7 relative above the rounding floor of the problem).  A
deviation of an opt-in inexact mode (fp32 Jacobians, CG at 1e-8) beyond 1e-6 is reported with the LM iteration at which the runs part, the SAME
problem re-run on the exact path (which must follow the oracle: otherwise HARD) and each approximation on its own."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# A run of more than LONG_RUN LM iterations has let the trust region grow to 1e8 .. 1e9 (outliers, barely determined problems): the damping is ~1e-12 of the
# diagonal, the reduced system has a condition number of ~1e12 and fp64 ROUNDING ORDER decides accept / reject decisions -- two correct implementations
# of the same loop end 1e-5 apart there (and so does the multi-threaded oracle against itself).  Mismatches of such runs are counted apart.
LONG_RUN = 40


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--big", action="store_true", help="include shapes above 213 cameras (streaming CG; the oracle takes seconds each)")
    ap.add_argument("--options", action="store_true", help="random LM options as well (the same on both sides): iteration limits 0 / 1 / 3 / 50, initial radius 1e-2 .. 1e12, "
                                                           "tolerances 0 .. 1e-3, Jacobi scaling off, a tight diagonal clamp, min_relative_decrease, max_radius")
    ap.add_argument("--only", type=int, default=-1, help="run this case of the sequence alone (the random stream is advanced through the others) and print its traces")
    args = ap.parse_args()
    import sfm_toy_library_amd as sfm
    from sfm_toy_library_amd import capi
    from oracle import oracle_py as oracle
    oracle.set_num_threads(1)          # (one thread: its sums are then in a fixed order -- on the chaotic cases of this sweep, eighty LM iterations on a barely
                                       # determined problem with outliers, the multi-threaded oracle itself ends on 50 ... 83 iterations from run to run)
    rng = np.random.default_rng(args.seed)
    cams_small = [1, 2, 3, 4, 5, 7, 8, 12, 20, 31, 32, 33, 42, 43, 44, 60, 90, 130]
    cams_big = [213, 214, 230, 300]
    hard = soft = inexact = chaotic = 0
    worst = {}
    t0 = time.time()
    for case in range(args.cases):
        n_cam = int(rng.choice(cams_big if (args.big and rng.random() < 0.25) else cams_small))
        banded = n_cam >= 32 and rng.random() < 0.3
        max_views = min(n_cam, 12)
        views = "banded" if banded else int(rng.integers(1, max_views + 1))
        n_pt = int(rng.choice([1, 3, 17, 64, 200, 777, 2000])) if n_cam < 200 else int(rng.choice([1500, 4000]))
        if banded:
            n_pt = max(n_pt, 40 * n_cam // 4)
        noise = float(rng.choice([0.0, 0.5, 2.0]))
        pseed = int(rng.integers(1, 1 << 30))
        prob = sfm.make_problem("cfg2", n_cam=n_cam, n_pt=n_pt, views=views, seed=pseed, noise_px=noise).copy()
        kind = rng.random()
        what = "plain"
        if kind < 0.25 and prob.n_obs > 20:                       # gross outliers: rejected steps, radius shrinks
            k = max(1, prob.n_obs // 50)
            idx = rng.choice(prob.n_obs, size=k, replace=False)
            prob.obs_xy[idx] += rng.normal(0.0, 80.0, size=(k, 2)).astype(np.float32)
            what = "outliers"
            if args.only == case:
                print("make_problem('cfg2', n_cam=%d, n_pt=%d, views=%r, seed=%d, noise_px=%r); outliers: idx=%r delta=%r" % (
                    n_cam, n_pt, views, pseed, noise, idx.tolist(), (prob.obs_xy[idx] - sfm.make_problem("cfg2", n_cam=n_cam, n_pt=n_pt, views=views, seed=pseed, noise_px=noise).obs_xy[idx]).tolist()))
        elif kind < 0.45:                                         # far from the minimum
            prob.cam6[1:, :3] += rng.normal(0.0, 0.04, size=(n_cam - 1, 3))
            prob.cam6[1:, 3:] += rng.normal(0.0, 0.08, size=(n_cam - 1, 3))
            prob.pt3 += rng.normal(0.0, 0.06, size=prob.pt3.shape)
            prob.cam6 = prob.cam6.astype(np.float32).astype(np.float64); prob.pt3 = prob.pt3.astype(np.float32).astype(np.float64)
            what = "far start"
        elif kind < 0.55:                                         # already converged: first step ends the run
            _c, _p, _f, _s, _ = oracle.solve(prob, sfm.SfmbaOptions.defaults(max_seconds=0.0))
            prob.cam6, prob.pt3, prob.focal = _c.copy(), _p.copy(), float(_f)
            what = "at the minimum"
        precision = int(rng.integers(0, 2))
        linear = int(rng.integers(0, 3))
        resident = rng.random() < 0.5
        if args.only >= 0 and case != args.only:
            continue
        if args.only >= 0:
            resident = True
        okw = {}
        if args.options:
            if rng.random() < 0.5: okw["max_iters"] = int(rng.choice([0, 1, 3, 50]))
            if rng.random() < 0.4: okw["initial_radius"] = float(rng.choice([1e-2, 1.0, 1e2, 1e8, 1e12]))
            if rng.random() < 0.3: okw["function_tolerance"] = float(rng.choice([0.0, 1e-12, 1e-3]))
            if rng.random() < 0.2: okw["gradient_tolerance"] = float(rng.choice([0.0, 1e-4, 1.0]))
            if rng.random() < 0.2: okw["parameter_tolerance"] = float(rng.choice([0.0, 1e-4]))
            if rng.random() < 0.2: okw["jacobi_scaling"] = 0
            if rng.random() < 0.15: okw["min_lm_diagonal"], okw["max_lm_diagonal"] = 1e-2, 1e2
            if rng.random() < 0.15: okw["min_relative_decrease"] = float(rng.choice([0.0, 0.25]))
            if rng.random() < 0.15: okw["max_radius"] = float(rng.choice([1e4, 1e6]))
            if rng.random() < 0.1: okw["max_consecutive_invalid_steps"] = 1
            if "function_tolerance" in okw and okw["function_tolerance"] == 0.0 and "max_iters" not in okw: okw["max_iters"] = 60       # (bounded)
        opt_o = sfm.SfmbaOptions.defaults(max_seconds=0.0, **okw)
        cam_o, pt_o, f_o, s_o, tr_o = oracle.solve(prob, opt_o)
        opt = capi.default_options(max_seconds=0.0, precision=precision, linear_solver=linear, **okw)
        try:
            if not resident:
                cam, pt, f, s, tr = capi.solve(prob, opt)
                how = "one-shot"
            else:
                with capi.Problem(prob, precision=precision) as P:
                    s1, tr1 = P.solve(opt)
                    P.reset()
                    s, tr = P.solve(opt)
                    cam, pt, f = P.get_params()
                    if s1["termination_name"] != s["termination_name"] or abs(s1["final_cost"] - s["final_cost"]) > 1e-5 * abs(s["final_cost"]) + 1e-13 * abs(s["initial_cost"]):
                        print("case %d: resident handle not repeatable after reset: %s it %d cost %r, then %s it %d cost %r" % (
                            case, s1["termination_name"], s1["iterations"], s1["final_cost"], s["termination_name"], s["iterations"], s["final_cost"]))
                        hard += 1
                    if args.only >= 0:
                        for name, t in (("first solve", tr1), ("second solve (after reset)", tr)):
                            print(name)
                            for r in t[:12] + (t[-6:] if len(t) > 18 else t[12:]):
                                print("   it %3d valid %d ok %d cost %.9e step %.3e rho %.3e radius %.3e lin_iters %d" % (
                                    r["iteration"], r["step_is_valid"], r["step_is_successful"], r["cost"], r["step_norm"], r["relative_decrease"], r["trust_region_radius"], r["linear_iters"]))
                        fd = next((k for k in range(min(len(tr1), len(tr))) if tr1[k]["cost"] != tr[k]["cost"] or tr1[k]["step_is_successful"] != tr[k]["step_is_successful"]), None)
                        print("   first LM iteration at which the two solves of the handle differ: %r" % (fd,))
                        if fd is not None:
                            for k in range(max(0, fd - 1), min(fd + 3, len(tr1), len(tr))):
                                print("      it %d: first %.15e ok %d radius %.6e | second %.15e ok %d radius %.6e" % (k, tr1[k]["cost"], tr1[k]["step_is_successful"], tr1[k]["trust_region_radius"],
                                                                                                                  tr[k]["cost"], tr[k]["step_is_successful"], tr[k]["trust_region_radius"]))
                        for rep in range(3):
                            P.reset()
                            sx, trx = P.solve(opt)
                            print("   same handle, reset, same options again: %s it %d cost %r" % (sx["termination_name"], sx["iterations"], sx["final_cost"]))
                        with capi.Problem(prob, precision=precision) as Q:
                            for rep in range(2):
                                sq, _ = Q.solve(opt)
                                print("   fresh handle, solve %d: %s it %d cost %r" % (rep, sq["termination_name"], sq["iterations"], sq["final_cost"]))
                                Q.reset()
                        for li in (0, 1, 2):
                            P.reset()
                            sx, _ = P.solve(capi.default_options(max_seconds=0.0, precision=precision, linear_solver=li))
                            print("   same handle, reset, linear %d: %s it %d cost %r message %r" % (li, sx["termination_name"], sx["iterations"], sx["final_cost"], sx.get("message")))
                how = "resident x2"
        except Exception as e:
            print("case %d: EXCEPTION %s: %s  (n_cam %d n_pt %d views %s %s precision %d linear %d)" % (case, type(e).__name__, e, n_cam, n_pt, views, what, precision, linear))
            hard += 1
            continue
        exact = precision == 0 and linear in (0, 2)
        bar = 1e-7 if exact else 1e-6          # (exact path: 2e-8 was the worst of 1 500 cases -- 92 LM iterations on an under-determined problem with outliers)
        scale = abs(s_o["final_cost"]) + 1e-14 * abs(s_o["initial_cost"]) + 1e-300
        rel = abs(s["final_cost"] - s_o["final_cost"]) / scale
        key = (precision, linear)
        worst[key] = max(worst.get(key, 0.0), rel)
        desc = "n_cam %d n_pt %d n_obs %d views %s noise %.1f %s precision %d linear %d %s%s" % (n_cam, prob.n_pt, prob.n_obs, views, noise, what, precision, linear, how, (" " + repr(okw)) if okw else "")
        if s["termination_name"] != s_o["termination_name"] or not (rel <= bar):
            # a cost inside the rounding floor of an exactly satisfiable problem compares against the initial cost
            floor = 1e-12 * abs(s_o["initial_cost"]) + prob.n_obs * np.sqrt(2.0 * max(s_o["final_cost"], 0.0) / max(prob.n_obs, 1)) * 1e-11
            if s["termination_name"] == s_o["termination_name"] and abs(s["final_cost"] - s_o["final_cost"]) <= floor:
                continue
            long_run = s_o["iterations"] > LONG_RUN and s["termination_name"] == s_o["termination_name"] and rel <= 1e-3
            print("case %d %s: %s | oracle %s it %d cost %.12e (initial %.3e) | hip %s it %d cost %.12e (rel %.2e)" % (
                case, ("long run" if long_run else "HARD") if exact else "inexact mode", desc, s_o["termination_name"], s_o["iterations"], s_o["final_cost"], s_o["initial_cost"],
                s["termination_name"], s["iterations"], s["final_cost"], rel))
            if exact and long_run:
                chaotic += 1
            elif exact:
                hard += 1
            else:
                inexact += 1
            # where do the two runs part?  first LM iteration whose cost differs by more than 1e-6 relative, or whose accept / reject decision differs
            first = next((k for k in range(min(len(tr), len(tr_o))) if tr[k]["step_is_successful"] != tr_o[k]["step_is_successful"]
                          or abs(tr[k]["cost"] - tr_o[k]["cost"]) > 1e-6 * abs(tr_o[k]["cost"])), None)
            if first is not None:
                print("        first difference at LM iteration %d: oracle cost %.9e ok %d radius %.3e | hip cost %.9e ok %d radius %.3e" % (
                    first, tr_o[first]["cost"], tr_o[first]["step_is_successful"], tr_o[first]["trust_region_radius"],
                    tr[first]["cost"], tr[first]["step_is_successful"], tr[first]["trust_region_radius"]))
            if not exact:
                # the same problem in the reference's own arithmetic (fp64, factorised): does THAT follow the oracle?
                c2, p2, f2, s2, tr2 = capi.solve(prob, capi.default_options(max_seconds=0.0, precision=0, linear_solver=0, **okw))
                rel2 = abs(s2["final_cost"] - s_o["final_cost"]) / scale
                off = s2["termination_name"] != s_o["termination_name"] or s2["iterations"] != s_o["iterations"] or (rel2 > 1e-7 and abs(s2["final_cost"] - s_o["final_cost"]) > floor)
                long2 = off and s_o["iterations"] > LONG_RUN and s2["termination_name"] == s_o["termination_name"] and rel2 <= 1e-3
                print("        fp64 + Cholesky on the same problem: %s it %d cost %.12e (rel %.2e)%s" % (
                    s2["termination_name"], s2["iterations"], s2["final_cost"], rel2, "  <-- also off (long run)" if long2 else "  <-- ALSO OFF: HARD" if off else ""))
                hard += 1 if (off and not long2) else 0
                chaotic += 1 if long2 else 0
                # which of the two approximations moves it: fp32 Jacobians with the factorisation, the fp64 CG at 1e-8
                for pr, li in ((1, 0), (0, 1)):
                    if (pr, li) == (precision, linear):
                        continue
                    s3 = capi.solve(prob, capi.default_options(max_seconds=0.0, precision=pr, linear_solver=li, **okw))[3]
                    print("        precision %d linear %d: %s it %d cost %.12e (rel %.2e)" % (pr, li, s3["termination_name"], s3["iterations"], s3["final_cost"],
                                                                                       abs(s3["final_cost"] - s_o["final_cost"]) / scale))
        elif s["iterations"] != s_o["iterations"] and exact and s_o["final_cost"] > 1e-9 * s_o["initial_cost"]:
            lr = s_o["iterations"] > LONG_RUN
            print("case %d %s (iteration count of the exact path): %s | iterations %d vs oracle %d, cost rel %.2e" % (case, "long run" if lr else "HARD", desc, s["iterations"], s_o["iterations"], rel))
            if lr:
                chaotic += 1
            else:
                hard += 1
        elif s["iterations"] != s_o["iterations"]:
            print("case %d soft: %s | iterations %d vs oracle %d, cost rel %.2e" % (case, desc, s["iterations"], s_o["iterations"], rel))
            soft += 1
    print("fuzz_parity: %d cases: %d HARD (the exact path -- fp64, factorised or AUTO -- off the oracle on a run of <= %d LM iterations, or an exception, or a termination that differs), "
          "%d exact-path differences on longer (chaotic) runs, %d deviations of the opt-in inexact modes (F32J / CG at 1e-8) beyond 1e-6, "
          "%d iteration-count-only differences, %.0f s; worst relative cost difference by (precision, linear): %s" % (
        args.cases, hard, LONG_RUN, chaotic, inexact, soft, time.time() - t0, {k: "%.1e" % v for k, v in sorted(worst.items())}))
    return 1 if hard else 0


if __name__ == "__main__":
    sys.exit(main())
