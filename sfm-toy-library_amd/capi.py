"""ctypes binding of the C ABI in include/sfmba.h (libsfmba_hip.so, built from csrc/).

This is the only way Python code (tests, bench.py, the sharded driver) reaches the product: through
the same extern "C" entry points the C++ shim in host/ calls.  If the shared library has not been
built, or no HIP device is present, calls fail loudly -- there is no CPU fallback.
"""
import ctypes as C
import os
import numpy as np

from .structs import SfmbaOptions, SfmbaSummary, SfmbaIteration

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libsfmba_hip.so")
if os.environ.get("SFMBA_LIB"):          # development aid: A/B a library built from another commit (tools/ab/) through the same binding
    LIB_PATH = os.path.abspath(os.environ["SFMBA_LIB"])
_lib = None

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)

# every extern "C" symbol include/sfmba.h declares (tests check the library exports all of them)
SYMBOLS = [
    "sfmba_options_default", "sfmba_abi_version", "sfmba_last_error", "sfmba_device_count", "sfmba_solve",
    "sfmba_problem_create", "sfmba_problem_reset", "sfmba_problem_set_params", "sfmba_problem_solve",
    "sfmba_problem_get_params", "sfmba_problem_destroy", "sfmba_problem_stream", "sfmba_problem_reduced_dim",
    "sfmba_problem_eval_residuals", "sfmba_problem_eval_jacobian", "sfmba_problem_build_reduced",
    "sfmba_dense_spd_solve", "sfmba_shard_begin", "sfmba_shard_reduce_len", "sfmba_shard_reduce_buf",
    "sfmba_shard_scalars_buf", "sfmba_shard_partial_build", "sfmba_shard_solve_update", "sfmba_shard_finish",
    "sfmba_shard_end", "sfmba_problem_set_profiling", "sfmba_problem_get_profile",
    "sfmba_problem_create_sharded", "sfmba_shard_setup_finish", "sfmba_shard_setup_len", "sfmba_shard_setup_buf", "sfmba_release_cache", "sfmba_triangulate",
    "sfmba_find_2d3d_matches", "sfmba_merge_candidates", "sfmba_problem_append",
    "sfmba_comm_unique_id", "sfmba_comm_create", "sfmba_comm_destroy", "sfmba_comm_allreduce", "sfmba_problem_solve_sharded",
    "sfmba_comm_allreduce_f32", "sfmba_problem_set_allreduce_f32", "sfmba_shard_last_exchange",
    "sfmba_problem_create_ex", "sfmba_comm_abort", "sfmba_comm_reduce_scatter", "sfmba_problem_set_reduce_scatter",
    "sfmba_comm_allgather", "sfmba_problem_set_allgather", "sfmba_comm_size", "sfmba_device_warmup",
]


def triangulate(K, P_left, P_right, left_xy, right_xy, max_reproj_px=10.0, device=0):
    """SfMStereoUtilities::triangulateViews for aligned matches on the GPU: (points3d [n,3] float32, keep [n] bool, err [n,2])."""
    K = np.ascontiguousarray(K, dtype=np.float32).reshape(9)
    Pl = np.ascontiguousarray(P_left, dtype=np.float32).reshape(12)
    Pr = np.ascontiguousarray(P_right, dtype=np.float32).reshape(12)
    l = np.ascontiguousarray(left_xy, dtype=np.float32).reshape(-1, 2)
    r = np.ascontiguousarray(right_xy, dtype=np.float32).reshape(-1, 2)
    n = l.shape[0]
    X = np.zeros((n, 3), dtype=np.float32); keep = np.zeros(n, dtype=np.uint8); err = np.zeros((n, 2), dtype=np.float32)
    fp = C.POINTER(C.c_float)
    _check(lib().sfmba_triangulate(C.c_int(device), C.c_int64(n), l.ctypes.data_as(fp), r.ctypes.data_as(fp), K.ctypes.data_as(fp),
                                   Pl.ctypes.data_as(fp), Pr.ctypes.data_as(fp), C.c_float(max_reproj_px), X.ctypes.data_as(fp),
                                   keep.ctypes.data_as(C.POINTER(C.c_ubyte)), err.ctypes.data_as(fp)))
    return X, keep.astype(bool), err


SFMBA_ERR_CAPACITY = 5


def _flat_views(views):
    """list of {view: feature} per cloud point -> CSR arrays in ascending view order (std::map iteration order)."""
    ptr = np.zeros(len(views) + 1, dtype=np.int64)
    vi, fi = [], []
    for i, m in enumerate(views):
        for v in sorted(m):
            vi.append(v)
            fi.append(m[v])
        ptr[i + 1] = len(vi)
    return ptr, np.asarray(vi, dtype=np.int32), np.asarray(fi, dtype=np.int32)


def _flat_matches(match_matrix):
    """{(left, right): [(query, train, distance), ...]} -> flattened pair arrays (any key order, every key kept)."""
    left, right, ptr, q, t, d = [], [], [0], [], [], []
    for (l, r), lst in match_matrix.items():
        left.append(l)
        right.append(r)
        for m in lst:
            q.append(m[0]); t.append(m[1]); d.append(m[2] if len(m) > 2 else 0.0)
        ptr.append(len(q))
    return (np.asarray(left, np.int32), np.asarray(right, np.int32), np.asarray(ptr, np.int64), np.asarray(q, np.int32),
            np.asarray(t, np.int32), np.asarray(d, np.float32))


def find_2d3d_matches(n_views, done_views, view_ptr, view_idx, feat_idx, pair_left, pair_right, pair_ptr, query_idx, train_idx,
                      cap=None, device=0):
    """sfmba_find_2d3d_matches on flat arrays: (out_ptr [n_views+1], out_point, out_feature)."""
    done = np.zeros(n_views, dtype=np.uint8)
    done[np.asarray(list(done_views), dtype=np.int64)] = 1
    view_ptr = np.ascontiguousarray(view_ptr, np.int64); view_idx = _i(view_idx); feat_idx = _i(feat_idx)
    pair_left = _i(pair_left); pair_right = _i(pair_right); pair_ptr = np.ascontiguousarray(pair_ptr, np.int64)
    query_idx = _i(query_idx); train_idx = _i(train_idx)
    n_pt = len(view_ptr) - 1
    cap = int(n_pt if cap is None else cap)
    lp = C.POINTER(C.c_int64)
    out_ptr = np.zeros(n_views + 1, dtype=np.int64)
    total = C.c_int64(0)
    for _ in range(2):
        op, of = np.zeros(max(cap, 1), np.int32), np.zeros(max(cap, 1), np.int32)
        rc = lib().sfmba_find_2d3d_matches(
            C.c_int(device), C.c_int(n_views), done.ctypes.data_as(C.POINTER(C.c_ubyte)), C.c_int(n_pt), _p(view_ptr, lp), _p(view_idx, _ip),
            _p(feat_idx, _ip), C.c_int(len(pair_left)), _p(pair_left, _ip), _p(pair_right, _ip), _p(pair_ptr, lp), _p(query_idx, _ip),
            _p(train_idx, _ip), _p(out_ptr, lp), _p(op, _ip), _p(of, _ip), C.c_int64(cap), C.byref(total))
        if rc != SFMBA_ERR_CAPACITY:
            break
        cap = int(total.value)
    _check(rc)
    return out_ptr, op[:total.value].copy(), of[:total.value].copy()


def merge_candidates(exist_xyz, new_xyz, max_dist=0.01, cap=None, device=0):
    """sfmba_merge_candidates: (cand_ptr [n_new+1], cand_idx) -- see include/sfmba.h."""
    ex = np.ascontiguousarray(exist_xyz, np.float32).reshape(-1, 3)
    nw = np.ascontiguousarray(new_xyz, np.float32).reshape(-1, 3)
    cap = int(4 * len(nw) + 1024 if cap is None else cap)
    lp, fp = C.POINTER(C.c_int64), C.POINTER(C.c_float)
    ptr = np.zeros(len(nw) + 1, dtype=np.int64)
    total = C.c_int64(0)
    for _ in range(2):
        idx = np.zeros(max(cap, 1), np.int32)
        rc = lib().sfmba_merge_candidates(C.c_int(device), C.c_int(len(ex)), _p(ex, fp), C.c_int(len(nw)), _p(nw, fp), C.c_float(max_dist),
                                          _p(ptr, lp), _p(idx, _ip), C.c_int64(cap), C.byref(total))
        if rc != SFMBA_ERR_CAPACITY:
            break
        cap = int(total.value)
    _check(rc)
    return ptr, idx[:total.value].copy()


def release_cache():
    """Return the device memory cached from destroyed problems to HIP; bytes released."""
    return int(lib().sfmba_release_cache())


class _KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("total_us", C.c_double), ("launches", C.c_int64)]


class SfmbaError(RuntimeError):
    pass


def _let_torch_open_the_device_first():
    """A torch wheel bundles its own HIP / HSA runtime beside the system one this library links against.  One process can hold both as
    long as torch's opens the device FIRST (every bench / sharded run does: torch.cuda.set_device comes before the first solve); the
    other way round torch.cuda then reports 'No HIP GPUs are available' -- e.g. a pytest selection that runs a plain C-ABI test before
    the first sharded one.  So: when torch is installed and sees a GPU, let it initialise before libsfmba_hip.so is loaded.  Nothing
    happens without torch or without a GPU; the C ABI itself never needs torch."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


def lib():
    global _lib
    if _lib is None:
        _let_torch_open_the_device_first()
        if not os.path.exists(LIB_PATH):
            raise SfmbaError("%s is missing: build it with `make -C %s` (or __graft_entry__.build()); "
                             "the MI355X back end has no CPU fallback" % (LIB_PATH, os.path.dirname(LIB_PATH)))
        L = C.CDLL(LIB_PATH)
        L.sfmba_last_error.restype = C.c_char_p
        L.sfmba_problem_stream.restype = C.c_void_p
        L.sfmba_shard_reduce_buf.restype = C.c_void_p
        L.sfmba_shard_scalars_buf.restype = C.c_void_p
        L.sfmba_shard_reduce_len.restype = C.c_int64
        L.sfmba_shard_setup_len.restype = C.c_int64
        L.sfmba_shard_setup_buf.restype = C.c_void_p
        L.sfmba_release_cache.restype = C.c_longlong
        for name in ("sfmba_problem_reset", "sfmba_problem_set_params", "sfmba_problem_solve", "sfmba_problem_get_params",
                     "sfmba_problem_destroy", "sfmba_problem_stream", "sfmba_problem_reduced_dim", "sfmba_problem_append",
                     "sfmba_problem_eval_residuals", "sfmba_problem_eval_jacobian", "sfmba_problem_build_reduced",
                     "sfmba_shard_begin", "sfmba_shard_reduce_len", "sfmba_shard_reduce_buf", "sfmba_shard_scalars_buf",
                     "sfmba_shard_partial_build", "sfmba_shard_solve_update", "sfmba_shard_finish", "sfmba_shard_end"):
            getattr(L, name).argtypes = None
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise SfmbaError("sfmba rc=%d: %s" % (rc, (lib().sfmba_last_error() or b"").decode()))


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a, t):
    return a.ctypes.data_as(t)


def device_count():
    return int(lib().sfmba_device_count())


def device_warmup(device=0, expected_obs=0):
    """sfmba_device_warmup: pay the process-wide first-call costs (HIP context, pinned pool, device chunks) now."""
    _check(lib().sfmba_device_warmup(C.c_int(device), C.c_int64(expected_obs)))


def default_options(**overrides):
    o = SfmbaOptions()
    lib().sfmba_options_default(C.byref(o))
    for k, v in overrides.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def _trace_rows(trace, n, cap):
    return [trace[i].as_dict() for i in range(min(n, cap))]


def solve(prob, opt=None, trace_cap=1024):
    """One-shot sfmba_solve on host arrays.  Returns (cam6, pt3, focal, summary, trace); prob untouched."""
    cam6, pt3 = _d(prob.cam6).copy(), _d(prob.pt3).copy()
    oc, op, oxy = _i(prob.obs_cam), _i(prob.obs_pt), _d(prob.obs_xy)
    focal = C.c_double(prob.focal)
    opt = opt or default_options()
    summ = SfmbaSummary()
    trace = (SfmbaIteration * trace_cap)()
    tl = C.c_int(0)
    _check(lib().sfmba_solve(C.c_int(prob.n_cam), _p(cam6, _dp), C.c_int(prob.n_pt), _p(pt3, _dp), C.c_int64(prob.n_obs),
                             _p(oc, _ip), _p(op, _ip), _p(oxy, _dp), C.byref(focal), C.byref(opt), C.byref(summ),
                             trace, C.c_int(trace_cap), C.byref(tl)))
    return cam6, pt3, focal.value, summ.as_dict(), _trace_rows(trace, tl.value, trace_cap)


def dense_spd_solve(A, b, method=0, tol=1e-12, max_iters=0, device=0):
    A, b = _d(A), _d(b)
    n = b.shape[0]
    x = np.zeros(n)
    info, iters = C.c_int(0), C.c_int(0)
    _check(lib().sfmba_dense_spd_solve(C.c_int(device), C.c_int(n), _p(A, _dp), _p(b, _dp), _p(x, _dp), C.c_int(method),
                                       C.c_double(tol), C.c_int(max_iters), C.byref(info), C.byref(iters)))
    return x, info.value, iters.value


class Problem:
    """Device-resident problem (sfmba_problem_*)."""

    def __init__(self, prob, precision=0, device=0, flags=0):
        """flags: SFMBA_CREATE_* (structs.CREATE_DETERMINISTIC); 0 goes through plain sfmba_problem_create."""
        self.n_cam, self.n_pt, self.n_obs = prob.n_cam, prob.n_pt, prob.n_obs
        cam6, pt3 = _d(prob.cam6), _d(prob.pt3)
        oc, op, oxy = _i(prob.obs_cam), _i(prob.obs_pt), _d(prob.obs_xy)
        self._h = C.c_void_p()
        self._template = (cam6.copy(), pt3.copy())
        if flags:
            _check(lib().sfmba_problem_create_ex(C.c_int(device), C.c_int(precision), C.c_int(flags), C.c_int(prob.n_cam), _p(cam6, _dp),
                                                 None, C.c_int(prob.n_pt), _p(pt3, _dp), C.c_int64(prob.n_obs), _p(oc, _ip), _p(op, _ip),
                                                 _p(oxy, _dp), C.c_double(prob.focal), C.c_int(0), C.c_int(1), C.byref(self._h)))
        else:
            _check(lib().sfmba_problem_create(C.c_int(device), C.c_int(precision), C.c_int(prob.n_cam), _p(cam6, _dp),
                                              C.c_int(prob.n_pt), _p(pt3, _dp), C.c_int64(prob.n_obs), _p(oc, _ip), _p(op, _ip),
                                              _p(oxy, _dp), C.c_double(prob.focal), C.byref(self._h)))

    def close(self):
        if self._h:
            lib().sfmba_problem_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def handle(self):
        return self._h

    @property
    def stream(self):
        return lib().sfmba_problem_stream(self._h)

    @property
    def reduced_dim(self):
        return int(lib().sfmba_problem_reduced_dim(self._h))

    def reset(self):
        _check(lib().sfmba_problem_reset(self._h))

    def set_params(self, cam6, pt3, focal):
        cam6, pt3 = _d(cam6), _d(pt3)
        _check(lib().sfmba_problem_set_params(self._h, _p(cam6, _dp), _p(pt3, _dp), C.c_double(focal)))

    def append(self, cam6, pt3, focal, obs_cam, obs_pt, obs_xy):
        """sfmba_problem_append: grow the resident problem by new observations (and, at the end of the arrays, new cameras /
        points); cam6 / pt3 are the FULL current parameter arrays."""
        cam6, pt3 = _d(cam6), _d(pt3)
        oc, op, oxy = _i(obs_cam), _i(obs_pt), _d(obs_xy)
        _check(lib().sfmba_problem_append(self._h, C.c_int(cam6.shape[0]), _p(cam6, _dp), C.c_int(pt3.shape[0]), _p(pt3, _dp),
                                          C.c_int64(len(oc)), _p(oc, _ip), _p(op, _ip), _p(oxy, _dp), C.c_double(focal)))
        self.n_cam, self.n_pt, self.n_obs = cam6.shape[0], pt3.shape[0], self.n_obs + len(oc)
        self._template = (cam6.copy(), pt3.copy())

    def get_params(self):
        cam6, pt3 = self._template[0].copy(), self._template[1].copy()
        focal = C.c_double(0.0)
        _check(lib().sfmba_problem_get_params(self._h, _p(cam6, _dp), _p(pt3, _dp), C.byref(focal)))
        return cam6, pt3, focal.value

    def solve(self, opt=None, trace_cap=1024):
        opt = opt or default_options()
        summ = SfmbaSummary()
        # the trace buffer is kept with the handle: allocating (and zeroing) 64 KB of ctypes array per call is measurable
        # next to a ~1 ms solve
        if getattr(self, "_trace_cap", 0) != trace_cap:
            self._trace = (SfmbaIteration * trace_cap)()
            self._trace_cap = trace_cap
        trace = self._trace
        tl = C.c_int(0)
        _check(lib().sfmba_problem_solve(self._h, C.byref(opt), C.byref(summ), trace, C.c_int(trace_cap), C.byref(tl)))
        return summ.as_dict(), _trace_rows(trace, tl.value, trace_cap)

    def set_profiling(self, enable):
        _check(lib().sfmba_problem_set_profiling(self._h, C.c_int(1 if enable else 0)))

    def get_profile(self):
        """{kernel: {total_us, launches, avg_us}} measured with HIP events on the solver's stream."""
        buf = (_KernelTime * 64)()
        n = C.c_int(0)
        _check(lib().sfmba_problem_get_profile(self._h, buf, C.c_int(64), C.byref(n)))
        out = {}
        for i in range(min(n.value, 64)):
            k = buf[i]
            out[k.name.decode()] = dict(total_us=k.total_us, launches=int(k.launches), avg_us=k.total_us / max(1, k.launches))
        return out

    def eval_residuals(self):
        res = np.zeros(2 * self.n_obs)
        cost = C.c_double(0.0)
        _check(lib().sfmba_problem_eval_residuals(self._h, _p(res, _dp), C.byref(cost)))
        return res.reshape(-1, 2), cost.value

    def eval_jacobian(self):
        n = self.n_obs
        jc, jp, jf = np.zeros(12 * n), np.zeros(6 * n), np.zeros(2 * n)
        _check(lib().sfmba_problem_eval_jacobian(self._h, _p(jc, _dp), _p(jp, _dp), _p(jf, _dp)))
        return jc.reshape(n, 2, 6), jp.reshape(n, 2, 3), jf.reshape(n, 2)

    def build_reduced(self, radius, opt=None):
        d = self.reduced_dim
        S, rhs, scale = np.zeros(d * d), np.zeros(d), np.zeros(d)
        opt = opt or default_options()
        _check(lib().sfmba_problem_build_reduced(self._h, C.byref(opt), C.c_double(radius), _p(S, _dp), _p(rhs, _dp), _p(scale, _dp)))
        return S.reshape(d, d), rhs, scale
