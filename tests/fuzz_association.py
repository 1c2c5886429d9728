"""Seed sweep of find2D3DMatches / mergeNewPointCloud on the MI355X against the oracle's restatement of the reference loops (SfM.cpp:471-629) -- the
scenarios of tests/test_gpu_association.py (duplicates, negative indices, ties, NaN / inf, hashed far cells, order-dependent mutations) over many more
seeds and sizes than the suite runs; every comparison EXACT.  Test infrastructure (calls the oracle); GPU box.

    python tests/fuzz_association.py [--seeds N] [--first S]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=150)
    ap.add_argument("--first", type=int, default=1000)
    args = ap.parse_args()
    from sfm_toy_library_amd import capi
    from oracle import association_oracle as ao
    import test_gpu_association as T
    bad = 0
    t0 = time.time()
    for seed in range(args.first, args.first + args.seeds):
        rng = np.random.default_rng(seed)
        try:
            # find2D3DMatches
            n_views = int(rng.integers(3, 14))
            n_done = int(rng.integers(1, max(2, n_views - 1)))
            done, cloud, mm, _ = T.scenario(seed, n_views=n_views, n_feat=int(rng.choice([5, 60, 400])), n_pt=int(rng.choice([1, 50, 300, 2000])),
                                            n_done=n_done, matches_per_pair=int(rng.choice([2, 90, 600])), junk=bool(rng.random() < 0.8))
            want = ao.find_2d3d_matches(n_views, done, cloud, mm)
            got, _ = T.run_find_capi(capi, n_views, done, cloud, mm, cap=(1 if rng.random() < 0.3 else None))
            if got != want:
                bad += 1; print("seed %d find2D3DMatches MISMATCH (views %d done %r)" % (seed, n_views, done))
            # merge candidates (radius join)
            ex, nw = T.clustered_points(rng, int(rng.choice([8, 400, 3000])), int(rng.choice([8, 300, 2000])))
            ptr_o, idx_o = ao.radius_candidates(ex, nw)
            ptr, idx = capi.merge_candidates(ex, nw, cap=(16 if rng.random() < 0.5 else None))
            if not (np.array_equal(ptr, ptr_o) and np.array_equal(idx, idx_o)):
                bad += 1; print("seed %d merge candidates MISMATCH (%d existing, %d new)" % (seed, len(ex), len(nw)))
            # mergeNewPointCloud through the reference signature
            cloud2, new2, mm2 = T.merge_scenario(seed)
            want2 = [(p.copy(), dict(v)) for p, v in cloud2]
            n_new, n_merged, pushed = ao.merge_new_point_cloud(want2, [(p.copy(), dict(v)) for p, v in new2], mm2)
            got2, g_new, g_merged, g_pushed = T.run_merge_shim(capi, 6, cloud2, new2, mm2)
            same = (g_new, g_merged) == (n_new, n_merged) and len(got2) == len(want2) and all(np.array_equal(gp, wp) and gv == wv for (gp, gv), (wp, wv) in zip(got2, want2))
            by_pair, g_by_pair = {}, {}
            for (l, r, pos) in pushed: by_pair.setdefault((l, r), []).append(mm2[(l, r)][pos][:2])
            for (l, r, q, t) in g_pushed: g_by_pair.setdefault((l, r), []).append((q, t))
            if not same or g_by_pair != by_pair:
                bad += 1; print("seed %d mergeNewPointCloud MISMATCH" % seed)
        except Exception as e:
            bad += 1; print("seed %d EXCEPTION %s: %s" % (seed, type(e).__name__, e))
    print("fuzz_association: %d seeds: %d mismatches, %.0f s" % (args.seeds, bad, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
