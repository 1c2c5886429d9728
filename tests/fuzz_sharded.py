"""Randomised sweep of the SHARDED solve (sfmba_problem_solve_sharded) on ONE rank against the oracle (test infrastructure -- it calls the oracle; GPU
box): the four forms of DESIGN.md section 6 -- replicated CG / exact solve on the all-reduced system, distributed CG on owned blocks, implicit Schur
product, block rows sharded -- run their whole exchange choreography with a world of one (every pack / transform / slice kernel executes; only the
collectives are the identity), on random shapes, both precisions, the three solver settings and the CG variants (plain block-Jacobi, fp64 exchange).

    python tests/fuzz_sharded.py [--cases N] [--seed S]"""
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
    ap.add_argument("--cases", type=int, default=120)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import sfm_toy_library_amd as sfm
    from sfm_toy_library_amd import capi
    from sfm_toy_library_amd.sharded import HipShardBackend, HipRowShardBackend, solve_sharded_native
    from oracle import oracle_py as oracle
    oracle.set_num_threads(1)
    rng = np.random.default_rng(args.seed)
    bad = 0
    forms = ["replicated", "distributed_cg", "implicit_schur", "row_sharded"]
    counts = dict.fromkeys(forms, 0)
    t0 = time.time()
    for case in range(args.cases):
        n_cam = int(rng.choice([1, 2, 3, 5, 8, 14, 25, 43, 44, 60, 90, 130]))
        banded = n_cam >= 32 and rng.random() < 0.25
        views = "banded" if banded else (1 if rng.random() < 0.08 else (2, min(max(n_cam, 2), int(rng.integers(2, 9)))))
        if n_cam == 1:
            views = 1
        n_pt = int(rng.choice([30, 150, 600, 1500]))
        if banded:
            n_pt = max(n_pt, 10 * n_cam)
        prob = sfm.make_problem("cfg2", n_cam=n_cam, n_pt=n_pt, views=views, seed=int(rng.integers(1, 1 << 30)), noise_px=float(rng.choice([0.0, 0.5, 2.0]))).copy()
        precision = int(rng.integers(0, 2))
        linear = int(rng.integers(0, 3))
        form = str(rng.choice(forms))
        counts[form] += 1
        okw = {}
        if form == "distributed_cg": okw["shard_distributed_cg"] = 1
        if form == "implicit_schur": okw["shard_distributed_cg"] = 2
        if rng.random() < 0.2: okw["pcg_coarse_space"] = -1
        if rng.random() < 0.2: okw["shard_f32_exchange"] = -1
        if rng.random() < 0.2: okw["shard_two_phase"] = int(rng.choice([-1, 1]))
        desc = "n_cam %d n_pt %d n_obs %d views %r precision %d linear %d %s %r" % (n_cam, n_pt, prob.n_obs, views, precision, linear, form, okw)
        try:
            s_o = oracle.solve(prob, sfm.SfmbaOptions.defaults(max_seconds=0.0))[3]
            opt = capi.default_options(max_seconds=0.0, precision=precision, linear_solver=linear, **okw)
            backend = (HipRowShardBackend if form == "row_sharded" else HipShardBackend)(prob, 0, 1, device=0, precision=precision)
            try:
                s = solve_sharded_native(backend, opt)
                backend.reset()
                s2 = solve_sharded_native(backend, opt)
            finally:
                backend.close()
        except Exception as e:
            bad += 1
            print("case %d EXCEPTION %s: %s | %s" % (case, type(e).__name__, e, desc))
            continue
        exact = precision == 0 and linear in (0, 2) and form in ("replicated", "row_sharded")
        scale = abs(s_o["final_cost"]) + 1e-14 * abs(s_o["initial_cost"]) + 1e-300
        rel = abs(s["final_cost"] - s_o["final_cost"]) / scale
        floor = 1e-12 * abs(s_o["initial_cost"]) + 1e-13
        tiny = s_o["final_cost"] <= 1e-9 * s_o["initial_cost"]
        off = s["termination_name"] != s_o["termination_name"]
        if not tiny and s_o["iterations"] <= LONG_RUN:
            off = off or (exact and s["iterations"] != s_o["iterations"]) or (rel > (1e-7 if exact else 1e-5) and abs(s["final_cost"] - s_o["final_cost"]) > floor)
            rep = abs(s2["final_cost"] - s["final_cost"]) / scale
            if s2["termination_name"] != s["termination_name"] or rep > 1e-5:
                bad += 1
                print("case %d NOT REPEATABLE after reset: %s | %s it %d cost %.12e, then %s it %d cost %.12e" % (
                    case, desc, s["termination_name"], s["iterations"], s["final_cost"], s2["termination_name"], s2["iterations"], s2["final_cost"]))
        if off:
            bad += 1
            print("case %d MISMATCH: %s | oracle %s it %d cost %.12e | hip %s it %d cost %.12e (rel %.2e)" % (
                case, desc, s_o["termination_name"], s_o["iterations"], s_o["final_cost"], s["termination_name"], s["iterations"], s["final_cost"], rel))
    print("fuzz_sharded: %d cases (%s): %d mismatches, %.0f s" % (args.cases, ", ".join("%s %d" % kv for kv in sorted(counts.items())), bad, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
