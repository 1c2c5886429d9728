"""F32J accuracy against the oracle on a few problems (max parameter difference, final cost, RMS), for the library selected by SFMBA_LIB.
   python tests/f32j_accuracy.py      (runs on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sfm_toy_library_amd as sfm
from sfm_toy_library_amd import capi
from oracle import oracle_py as oracle
oracle.build()
cases = [("tiny", dict(name="tiny")), ("small", dict(name="small")), ("cfg2", dict(name="cfg2")), ("cfg2_s7", dict(name="cfg2", seed=7)),
         ("mid", dict(name="cfg3", n_cam=60, n_pt=8003, seed=5)), ("wide", dict(name="cfg3", n_cam=230, n_pt=6001, seed=78)),
         ("few", dict(name="cfg2", n_cam=4, n_pt=400, views=3, seed=3)), ("banded", dict(name="cfg3_banded", n_cam=60, n_pt=6000))]
for tag, kw in cases:
    prob = sfm.make_problem(**kw)
    co, po, fo, so, tro = oracle.solve(prob, sfm.SfmbaOptions.defaults(max_seconds=0.0))
    for lin in (1, 2):
        cam, pt, f, s, tr = capi.solve(prob, capi.default_options(max_seconds=0.0, precision=1, linear_solver=lin))
        rms = lambda c: np.sqrt(2 * c / prob.n_obs)
        print("%-8s lin %d  it %d/%d  cam %.2e  pt %.2e  f %.2e  cost rel %.2e  rms diff %.2e px   rho-trace %s" % (
            tag, lin, s["iterations"], so["iterations"], np.abs(cam - co).max(), np.abs(pt - po).max(), abs(f - fo) / fo,
            abs(s["final_cost"] - so["final_cost"]) / so["final_cost"], abs(rms(s["final_cost"]) - rms(so["final_cost"])),
            " ".join("%.4f" % r["relative_decrease"] for r in tr[1:4])))
