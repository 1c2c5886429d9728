"""Scratch driver for GPU bring-up: traces GPU vs oracle, solve timings. Output -> gpurun_out/debug.json"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sfm_toy_library_amd as sfm
from sfm_toy_library_amd import capi

out = {}


def run(name, precision, linear=0, reps=3, compare=False, **kw):
    prob = sfm.make_problem(name, **kw)
    opt = capi.default_options(max_seconds=0.0, precision=precision, linear_solver=linear)
    t0 = time.time()
    P = capi.Problem(prob, precision=precision)
    t_create = time.time() - t0
    times = []
    for r in range(reps):
        P.reset()
        t0 = time.time()
        s, tr = P.solve(opt)
        times.append(time.time() - t0)
    key = "%s_p%d_l%d" % (name, precision, linear)
    out[key] = dict(create_s=t_create, solve_s=times, summary=s, trace=tr)
    print(key, "create %.3fs" % t_create, "solve", ["%.4f" % t for t in times], s["termination_name"], s["iterations"],
          "cost %.9e" % s["final_cost"], "lin_iters", s["linear_iters"], flush=True)
    if compare:
        from oracle import oracle_py as oracle
        t0 = time.time()
        cam_o, pt_o, f_o, s_o, tr_o = oracle.solve(prob, sfm.SfmbaOptions.defaults(max_seconds=0.0))
        print("   oracle %.3fs" % (time.time() - t0), s_o["termination_name"], s_o["iterations"], "cost %.9e" % s_o["final_cost"])
        for a, b in zip(tr, tr_o):
            print("   it %d gpu cost %.12e rho %.4e rad %.3e | oracle cost %.12e rho %.4e rad %.3e" %
                  (a["iteration"], a["cost"], a["relative_decrease"], a["trust_region_radius"], b["cost"], b["relative_decrease"], b["trust_region_radius"]))
        out[key]["oracle"] = s_o
    P.close()


if __name__ == "__main__":
    which = sys.argv[1:] or ["small", "cfg2", "cfg3"]
    if "small" in which:
        run("small", 0, compare=True)
    if "cfg2" in which:
        run("cfg2", 0, compare=True)
        run("cfg2", 1)
        run("cfg2", 0, linear=1)
    if "cfg3" in which:
        run("cfg3", 1)
        run("cfg3", 0)
        run("cfg3", 1, linear=1)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "debug.json"), "w") as f:
        json.dump(out, f, indent=1, default=str)
