"""The algebra behind the factored pair pass (csrc/sfmba_device.h: obs_factored / pair_product_factored), in numpy.

The camera block of an observation (A = [A_proj G | A_proj], G = -R [X]x K' with K' = (w w^T + (R^T - I) [w]x) / theta^2, or
G = -[X]x on the first-order branch theta^2 <= eps) factors as

    A = P [ -[X_g]x | I ] diag(Q, I),      X_g = R X,  Q = R K'      (X_g = X, Q = I on the first-order branch)

because R [X]x = [R X]x R for a rotation.  diag(Q, I) depends on the camera only, so a 6x6 block of the reduced camera matrix,
sum over the common points of A_a^T (C_a C_b^T) A_b, is E_a^T [ sum G_a^T N G_b ] E_b with N = P_a^T (C_a C_b^T) P_b -- what the
kernel sums per pair and transforms once per block.  CPU-only: no GPU, no library."""
import numpy as np


def skew(v):
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def camera(w):
    """R, K', first-order flag -- the camera table row (ba_kernels.hip: make_cam_table)."""
    th2 = float(w @ w)
    if th2 > np.finfo(float).eps:
        th = np.sqrt(th2)
        k = w / th
        R = np.cos(th) * np.eye(3) + np.sin(th) * skew(k) + (1.0 - np.cos(th)) * np.outer(k, k)
        K = (np.outer(w, w) + (R.T - np.eye(3)) @ skew(w)) / th2
        return R, K, False
    return np.eye(3) + skew(w), np.eye(3), True


def blocks(w, t, X, f):
    """Unfactored camera block A (2x6), projection Jacobian P (2x3), point block B = P R, and the factors G (3x6), E (6x6)."""
    R, K, first_order = camera(w)
    p = R @ X + t
    xp, yp, fz = p[0] / p[2], p[1] / p[2], f / p[2]
    P = fz * np.array([[1.0, 0.0, -xp], [0.0, 1.0, -yp]])
    B = P @ R
    Aw = -P @ skew(X) if first_order else -B @ skew(X) @ K
    A = np.hstack([Aw, P])
    Xg = X if first_order else R @ X
    Q = np.eye(3) if first_order else R @ K
    G = np.hstack([-skew(Xg), np.eye(3)])
    E = np.block([[Q, np.zeros((3, 3))], [np.zeros((3, 3)), np.eye(3)]])
    return A, P, B, G, E


def test_camera_block_factors():
    rng = np.random.default_rng(7)
    for trial in range(200):
        w = rng.normal(size=3) * (1e-9 if trial % 10 == 0 else 10.0 ** rng.uniform(-6, 0.4))      # includes the first-order branch
        if trial % 25 == 0:
            w = np.zeros(3)
        t = rng.normal(size=3) + np.array([0.0, 0.0, 5.0])
        X = rng.uniform(-1.0, 1.0, size=3)
        A, P, B, G, E = blocks(w, t, X, 2500.0)
        assert np.allclose(A, P @ G @ E, rtol=1e-11, atol=1e-9 * np.abs(A).max()), trial


def test_block_sum_factors():
    """sum_pairs A_a^T (C_a C_b^T) A_b == E_a^T [ sum_pairs G_a^T N G_b ] E_b,  N = P_a^T (C_a C_b^T) P_b,  C = B L."""
    rng = np.random.default_rng(11)
    for first_order_a in (False, True):
        wa = np.zeros(3) if first_order_a else rng.normal(size=3) * 0.3
        wb = rng.normal(size=3) * 0.7
        ta, tb = np.array([0.1, -0.2, 5.0]), np.array([-0.3, 0.1, 4.5])
        S = np.zeros((6, 6))
        Sf = np.zeros((6, 6))
        Ea = Eb = None
        for _ in range(40):
            X = rng.uniform(-1.0, 1.0, size=3)
            L = np.tril(rng.normal(size=(3, 3))) * 1e-3           # L^-1 diag(s_p) of the point (any lower-triangular matrix will do)
            Aa, Pa, Ba, Ga, Ea = blocks(wa, ta, X, 2500.0)
            Ab, Pb, Bb, Gb, Eb = blocks(wb, tb, X, 2500.0)
            Ca, Cb = Ba @ L.T, Bb @ L.T
            m = Ca @ Cb.T
            S += Aa.T @ m @ Ab
            Sf += Ga.T @ (Pa.T @ m @ Pb) @ Gb
        assert np.allclose(S, Ea.T @ Sf @ Eb, rtol=1e-10, atol=1e-12 * np.abs(S).max())


def test_pair_product_expansion():
    """The cross-product form the kernel evaluates: G_a^T N G_b = [[-[Xa]x N [Xb]x, [Xa]x N], [-N [Xb]x, N]]."""
    rng = np.random.default_rng(3)
    Xa, Xb, N = rng.normal(size=3), rng.normal(size=3), rng.normal(size=(3, 3))
    Ga, Gb = np.hstack([-skew(Xa), np.eye(3)]), np.hstack([-skew(Xb), np.eye(3)])
    want = Ga.T @ N @ Gb
    T = np.stack([np.cross(Xa, N[:, c]) for c in range(3)], axis=1)          # [Xa]x N, column by column
    got = np.zeros((6, 6))
    for r in range(3):
        got[r, 0:3] = np.cross(Xb, T[r])             # -(T[r] [Xb]x) = Xb x T[r]
        got[r, 3:6] = T[r]
        got[3 + r, 0:3] = np.cross(Xb, N[r])
        got[3 + r, 3:6] = N[r]
    assert np.allclose(got, want, rtol=1e-13, atol=1e-13)
