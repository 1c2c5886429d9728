"""Seeded synthetic bundle-adjustment problems (generator spec: SURVEY.md 8(d)).

The reference ships no problem generator; BASELINE.json's configs are stated on synthetic
random-pose / random-point problems, so this module IS the workload definition:

  points    uniform in [-1,1]^3
  cameras   centres on a shell of radius 4..6 around the origin, optical axis towards the origin
            (+ N(0,0.05 rad) jitter), random roll; world->camera pose (R,t), p = R X + t
            (BA.cpp:67-74).  Camera 0 is R=I, t=(0,0,5): it exercises the theta=0 branch of
            AngleAxisRotatePoint exactly like the reference's first view (Stereo.cpp:104, P=[I|0]).
  intrinsics f*=2500, 1024x768, c=(512,384) (SfM.cpp:70-72 with the Crazy Horse image size)
  observations exact projection + N(0,0.5 px), principal point subtracted (BA.cpp:151-153),
            rounded to fp32 (Point2f) and widened
  visibility each point seen by m distinct cameras drawn uniformly (p_z <= 0.1 rejected)
  initial guess  w += N(0,0.01), t += N(0,0.02), X += N(0,0.02), f0 = 1.02 f*; cameras and
            points rounded through fp32 (Matx34f / Point3f inputs, SfMCommon.h:84,99)
Observation order is point-major with ascending view index inside a point, which is the order
adjustBundle() adds residual blocks in (BA.cpp:142-166, std::map iteration).
"""
from dataclasses import dataclass, field
import numpy as np

F_TRUE = 2500.0
IMAGE_SIZE = (1024, 768)
PRINCIPAL_POINT = (512.0, 384.0)

# name -> (n_cam, n_pt, views_per_point, seed).  cfg numbers follow BASELINE.md section 3.
CONFIGS = {
    "crazyhorse_like": dict(n_cam=7, n_pt=1500, views=(2, 7), seed=1001),      # cfg 1 stand-in
    "cfg2": dict(n_cam=20, n_pt=5000, views=6, seed=1002),                       # 30 000 obs, fp64 parity
    "cfg3": dict(n_cam=200, n_pt=100000, views=10, seed=1003),                   # 1 000 000 obs, headline
    "cfg4": dict(n_cam=25, n_pt=12500, views=10, seed=1004),                     # one of 8 replicas (sub=g)
    "cfg5": dict(n_cam=1000, n_pt=500000, views=10, seed=1005),                  # 5 000 000 obs (10/pt assumed)
    # realistic co-visibility (reference shape: the incremental loop SfM.cpp:366-469 -- every new view sees its
    # neighbours): 200 cameras on a closed path around the scene, every point seen by a RUN of 2..30 NEIGHBOURING cameras
    # (power-law track length, mean ~10 => ~1M observations) => banded reduced system with a few very heavy blocks
    "cfg3_banded": dict(n_cam=200, n_pt=100000, views="banded", seed=1013),
    "banded_small": dict(n_cam=40, n_pt=3000, views="banded", seed=1014),        # the same shape at a size the CPU tests afford
    "tiny": dict(n_cam=4, n_pt=40, views=(2, 4), seed=999),                      # unit-test size
    "small": dict(n_cam=8, n_pt=400, views=(2, 6), seed=998),
}


@dataclass
class BAProblem:
    cam6: np.ndarray       # [n_cam, 6] float64  (angle-axis, translation) initial guess
    pt3: np.ndarray        # [n_pt, 3]  float64
    focal: float
    obs_cam: np.ndarray    # [n_obs] int32
    obs_pt: np.ndarray     # [n_obs] int32
    obs_xy: np.ndarray     # [n_obs, 2] float64, principal point subtracted
    cam6_true: np.ndarray = None
    pt3_true: np.ndarray = None
    focal_true: float = F_TRUE
    meta: dict = field(default_factory=dict)

    @property
    def n_cam(self):
        return int(self.cam6.shape[0])

    @property
    def n_pt(self):
        return int(self.pt3.shape[0])

    @property
    def n_obs(self):
        return int(self.obs_cam.shape[0])

    def copy(self):
        return BAProblem(self.cam6.copy(), self.pt3.copy(), float(self.focal), self.obs_cam.copy(),
                         self.obs_pt.copy(), self.obs_xy.copy(),
                         None if self.cam6_true is None else self.cam6_true.copy(),
                         None if self.pt3_true is None else self.pt3_true.copy(),
                         self.focal_true, dict(self.meta))

    def shard_points(self, rank, world):
        """Point-sharded sub-problem for rank `rank` of `world`: a contiguous block of
        points with all their observations; every camera and the focal are replicated."""
        lo = (self.n_pt * rank) // world
        hi = (self.n_pt * (rank + 1)) // world
        sel = (self.obs_pt >= lo) & (self.obs_pt < hi)
        sub = BAProblem(self.cam6.copy(), self.pt3[lo:hi].copy(), float(self.focal),
                        self.obs_cam[sel].copy(), (self.obs_pt[sel] - lo).astype(np.int32),
                        self.obs_xy[sel].copy(), None, None, self.focal_true,
                        dict(self.meta, shard=(rank, world), point_range=(lo, hi)))
        return sub


def rotvec_to_matrix(w):
    """Rodrigues, vectorised: w [...,3] -> R [...,3,3] (exact formula, theta=0 -> I)."""
    w = np.asarray(w, dtype=np.float64)
    theta = np.linalg.norm(w, axis=-1)
    safe = np.where(theta > 0, theta, 1.0)
    k = w / safe[..., None]
    K = np.zeros(w.shape[:-1] + (3, 3))
    K[..., 0, 1] = -k[..., 2]; K[..., 0, 2] = k[..., 1]
    K[..., 1, 0] = k[..., 2];  K[..., 1, 2] = -k[..., 0]
    K[..., 2, 0] = -k[..., 1]; K[..., 2, 1] = k[..., 0]
    s = np.sin(theta)[..., None, None]
    c = np.cos(theta)[..., None, None]
    eye = np.broadcast_to(np.eye(3), K.shape)
    return eye + s * K + (1.0 - c) * (K @ K)


def matrix_to_rotvec(R):
    from scipy.spatial.transform import Rotation
    return Rotation.from_matrix(np.asarray(R)).as_rotvec()


def project(cam6, pt3, focal, obs_cam, obs_pt):
    """Reference camera model (BA.cpp:67-86) without the observation term, vectorised numpy."""
    R = rotvec_to_matrix(cam6[:, :3])
    p = np.einsum("nij,nj->ni", R[obs_cam], pt3[obs_pt]) + cam6[obs_cam, 3:]
    return focal * p[:, :2] / p[:, 2:3], p[:, 2]


def _choose_views(rng, n_pt, n_cam, views):
    """Per point: m distinct cameras, ascending.  Returns (obs_pt, obs_cam)."""
    if isinstance(views, (tuple, list)):
        lo, hi = views
        m = rng.integers(lo, min(hi, n_cam) + 1, size=n_pt)
    else:
        m = np.full(n_pt, min(int(views), n_cam), dtype=np.int64)
    mmax = int(m.max())
    cams = np.empty((n_pt, mmax), dtype=np.int32)
    chunk = 16384
    for s in range(0, n_pt, chunk):
        e = min(n_pt, s + chunk)
        keys = rng.random((e - s, n_cam), dtype=np.float32)
        cams[s:e] = np.argpartition(keys, mmax - 1, axis=1)[:, :mmax].astype(np.int32)
    keep = np.arange(mmax)[None, :] < m[:, None]
    big = np.where(keep, cams, np.int32(2**30))
    big.sort(axis=1)
    obs_pt = np.repeat(np.arange(n_pt, dtype=np.int32), m)
    obs_cam = big[keep]
    return obs_pt, obs_cam.astype(np.int32)


def _banded_views(rng, n_pt, n_cam, target_mean=10.0, lo=2, hi=30):
    """Per point: a run of `m` CONSECUTIVE cameras of the (closed) camera path, m in [lo, hi] drawn from a truncated power law
    P(m) ~ m^-a whose exponent is chosen so that the mean track length is `target_mean` (real SfM tracks: many short, few
    long).  Returns (obs_pt, obs_cam) point-major, ascending camera inside a point."""
    hi = min(hi, n_cam)
    ms = np.arange(lo, hi + 1, dtype=np.float64)
    a_lo, a_hi = -3.0, 6.0
    for _ in range(60):                                   # bisection on the exponent (mean is monotone decreasing in a)
        a = 0.5 * (a_lo + a_hi)
        w = ms ** (-a)
        mean = float((w * ms).sum() / w.sum())
        if mean > target_mean:
            a_lo = a
        else:
            a_hi = a
    w = ms ** (-0.5 * (a_lo + a_hi))
    m = rng.choice(ms.astype(np.int64), size=n_pt, p=w / w.sum())
    start = rng.integers(0, n_cam, size=n_pt)
    mmax = int(m.max())
    cams = (start[:, None] + np.arange(mmax)[None, :]) % n_cam
    keep = np.arange(mmax)[None, :] < m[:, None]
    big = np.where(keep, cams, 2**30).astype(np.int64)
    big.sort(axis=1)
    obs_pt = np.repeat(np.arange(n_pt, dtype=np.int32), m)
    return obs_pt, big[keep].astype(np.int32)


def make_problem(name="cfg2", sub=None, n_cam=None, n_pt=None, views=None, seed=None,
                 noise_px=0.5, perturb=True):
    """Build one of CONFIGS (optionally overriding sizes).  `sub` selects one of the independent
    sub-problems of cfg4 (seed sequence [seed, sub])."""
    cfg = dict(CONFIGS[name]) if name in CONFIGS else {}
    n_cam = int(n_cam if n_cam is not None else cfg["n_cam"])
    n_pt = int(n_pt if n_pt is not None else cfg["n_pt"])
    views = views if views is not None else cfg["views"]
    seed = int(seed if seed is not None else cfg["seed"])
    rng = np.random.default_rng([seed, int(sub)] if sub is not None else seed)

    pt_true = rng.uniform(-1.0, 1.0, size=(n_pt, 3))
    banded = isinstance(views, str) and views == "banded"

    # camera centres on a shell, looking at the origin with jitter and random roll
    d = rng.normal(size=(n_cam, 3))
    if banded:
        # cameras along a closed, gently undulating path around the scene (consecutive cameras are neighbours in space too)
        ang = 2.0 * np.pi * (np.arange(n_cam) + rng.uniform(-0.2, 0.2, size=n_cam)) / n_cam
        d = np.stack([np.cos(ang), np.sin(ang), 0.25 * np.sin(3.0 * ang) + rng.normal(0.0, 0.03, size=n_cam)], axis=1)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    centre = d * rng.uniform(4.0, 6.0, size=(n_cam, 1))
    z = -centre / np.linalg.norm(centre, axis=1, keepdims=True)
    helper = np.where(np.abs(z[:, 2:3]) < 0.9, np.array([[0.0, 0.0, 1.0]]), np.array([[1.0, 0.0, 0.0]]))
    x = np.cross(helper, z)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    y = np.cross(z, x)
    R_look = np.stack([x, y, z], axis=1)                      # rows = camera axes in world coords
    roll = rng.uniform(0.0, 2.0 * np.pi, size=n_cam)
    R_roll = rotvec_to_matrix(np.stack([np.zeros(n_cam), np.zeros(n_cam), roll], axis=1))
    R_jit = rotvec_to_matrix(rng.normal(0.0, 0.05, size=(n_cam, 3)))
    R = R_jit @ R_roll @ R_look
    t = -np.einsum("nij,nj->ni", R, centre)
    R[0] = np.eye(3)
    t[0] = (0.0, 0.0, 5.0)
    w_true = matrix_to_rotvec(R)
    w_true[0] = 0.0
    cam_true = np.concatenate([w_true, t], axis=1)

    obs_pt, obs_cam = _banded_views(rng, n_pt, n_cam) if banded else _choose_views(rng, n_pt, n_cam, views)
    uv, pz = project(cam_true, pt_true, F_TRUE, obs_cam, obs_pt)
    good = pz > 0.1
    if not np.all(good):                                       # never happens for this geometry; keep the rule
        obs_pt, obs_cam, uv = obs_pt[good], obs_cam[good], uv[good]
    uv = uv + rng.normal(0.0, noise_px, size=uv.shape)
    obs_xy = uv.astype(np.float32).astype(np.float64)

    cam0 = cam_true.copy()
    pt0 = pt_true.copy()
    f0 = F_TRUE
    if perturb:
        dw = rng.normal(0.0, 0.01, size=(n_cam, 3))
        dt = rng.normal(0.0, 0.02, size=(n_cam, 3))
        dw[0] = 0.0
        dt[0] = 0.0
        cam0[:, :3] += dw
        cam0[:, 3:] += dt
        pt0 += rng.normal(0.0, 0.02, size=pt0.shape)
        f0 = 1.02 * F_TRUE
    cam0 = cam0.astype(np.float32).astype(np.float64)
    pt0 = pt0.astype(np.float32).astype(np.float64)

    return BAProblem(np.ascontiguousarray(cam0), np.ascontiguousarray(pt0), float(f0),
                     np.ascontiguousarray(obs_cam), np.ascontiguousarray(obs_pt),
                     np.ascontiguousarray(obs_xy), cam_true, pt_true, F_TRUE,
                     dict(name=name, sub=sub, seed=seed, noise_px=noise_px,
                          image_size=IMAGE_SIZE, principal_point=PRINCIPAL_POINT))
