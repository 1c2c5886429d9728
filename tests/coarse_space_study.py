"""CPU study (numpy + the oracle's reduced system): how many CG iterations does the block-Jacobi-preconditioned reduced system of a
realistic-co-visibility problem need with (a) no coarse space, (b) the 8 global gauge vectors of the product, (c) the same gauge vectors
restricted to G contiguous groups of cameras (7 G + 1 vectors: piecewise similarity transforms, the near-null space of a camera chain)?
    python tests/coarse_space_study.py [workload] [radius]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sfm_toy_library_amd as sfm
from oracle import oracle_py as oracle

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3_banded"
radius = float(sys.argv[2]) if len(sys.argv) > 2 else 1e4
kw = {}
if name == "cfg3_banded_small":
    name, kw = "cfg3_banded", dict(n_pt=20000)
prob = sfm.make_problem(name, **kw)
S, rhs, scale, info = oracle.build_reduced(prob, radius)
nc = prob.n_cam
d = 6 * nc + 1
assert S.shape == (d, d) and info == 0
# block-Jacobi: Lb Lb^T = diag blocks
Lb = np.zeros_like(S)
for j in range(nc):
    sl = slice(6 * j, 6 * j + 6)
    Lb[sl, sl] = np.linalg.cholesky(S[sl, sl])
Lb[d - 1, d - 1] = np.sqrt(S[d - 1, d - 1])
Li = np.linalg.inv(Lb)
St = Li @ S @ Li.T
bt = Li @ rhs


def skew(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])


def gauge(cam6, focal):
    """[8][d] gauge vectors in PARAMETER space (unscaled)."""
    R = sfm.synthetic.rotvec_to_matrix(cam6[:, :3])
    W = np.zeros((8, d))
    for j in range(nc):
        w = cam6[j, :3]; th2 = w @ w
        K = skew(w)
        if th2 > 1e-8:
            th = np.sqrt(th2); cq = 1 / th2 - (1 + np.cos(th)) / (2 * th * np.sin(th))
        else:
            cq = 1 / 12
        Ji = np.eye(3) + 0.5 * K + cq * K @ K
        for k in range(3):
            W[k, 6 * j + 3:6 * j + 6] = -R[j][:, k]
            W[3 + k, 6 * j:6 * j + 3] = -Ji[:, k]
        W[6, 6 * j + 3:6 * j + 6] = cam6[j, 3:]
        W[7, 6 * j + 5] = cam6[j, 5]
    W[7, d - 1] = focal
    return W


Wp = gauge(prob.cam6, prob.focal)
# scaled unknowns x_s = x / scale ; transformed x~ = Lb^T x_s
Wt_global = (Lb.T @ (Wp / scale).T).T


def pcg(Wt, tol=1e-8, maxit=2000):
    if Wt is not None:
        # drop dependent vectors
        AW = St @ Wt.T
        E = Wt @ AW
        Einv = np.linalg.pinv(E, rcond=1e-12)
    x = np.zeros(d); r = bt.copy()
    def prec(r):
        return r if Wt is None else r + Wt.T @ (Einv @ (Wt @ r))
    z = prec(r); p = z.copy(); rz = r @ z; r0 = np.sqrt(r @ r)
    for it in range(1, maxit + 1):
        q = St @ p
        a = rz / (p @ q)
        x += a * p; r -= a * q
        if np.sqrt(r @ r) <= tol * r0:
            return it
        z = prec(r); rz2 = r @ z
        p = z + (rz2 / rz) * p; rz = rz2
    return maxit


ev = np.linalg.eigvalsh(St)
print("%s: d = %d, radius %g; spectrum of the block-Jacobi preconditioned matrix: min %.2e, 10th %.2e, 30th %.2e, 60th %.2e, max %.2f" %
      (name, d, radius, ev[0], ev[9], ev[29], ev[59], ev[-1]))
print("CG iterations to 1e-8: no coarse space %d, 8 global gauge vectors %d" % (pcg(None), pcg(Wt_global)))
for G in (2, 4, 8, 16, 25):
    rows = []
    edges = np.linspace(0, nc, G + 1).astype(int)
    for g in range(G):
        mask = np.zeros(d)
        mask[6 * edges[g]:6 * edges[g + 1]] = 1.0
        for k in range(7):
            rows.append(Wt_global[k] * mask)
    rows.append(Wt_global[7])
    Wg = np.array(rows)
    print("  %2d groups of cameras (%3d coarse vectors): %d iterations" % (G, len(rows), pcg(Wg)))
# smooth partition of unity (hat functions along the camera path) instead of indicator functions
for G in (4, 8, 16):
    rows = []
    centres = np.linspace(0, nc, G, endpoint=False)
    idx = np.arange(nc)
    for g in range(G):
        dist = np.abs(((idx - centres[g] + nc / 2) % nc) - nc / 2)          # cyclic distance
        hat = np.clip(1.0 - dist / (nc / G), 0.0, 1.0)
        mask = np.zeros(d); mask[:6 * nc] = np.repeat(hat, 6)
        for k in range(7):
            rows.append(Wt_global[k] * mask)
    rows.append(Wt_global[7])
    print("  %2d hat functions on the cyclic path (%3d coarse vectors): %d iterations" % (G, len(rows), pcg(np.array(rows))))
