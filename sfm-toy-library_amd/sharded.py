"""One-process-per-GPU bundle adjustment of ONE problem whose points are sharded across ranks
(BASELINE.json config 5).

Points are conditionally independent given the cameras, so each rank eliminates its own points and the
only data-path exchange per LM iteration is the sum of the partial reduced camera systems
(torch.distributed all_reduce == RCCL over xGMI on the GPU box, gloo in the CPU tests).  Every rank then
solves the reduced system redundantly and takes the same accept/reject decision -- no broadcast.

The choreography (``solve_sharded``) is backend-agnostic: the product backend is ``HipShardBackend``
(the C ABI of include/sfmba.h on the rank's GPU); tests drive the same choreography with a CPU backend.
"""
import ctypes as C

import numpy as np

from . import capi
from .structs import SfmbaSummary


class _DevicePtr:
    """Exposes a raw device pointer through __cuda_array_interface__ so torch can wrap it without a copy."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


class _DevicePtrT:
    """The same for any element type (typestr '<f8' / '<f4')."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class HipShardBackend:
    """Rank-local half of the sharded solve on one MI355X."""

    def __init__(self, full_prob, rank, world, device=0, precision=1, flags=0):
        import torch
        self.torch = torch
        self.rank, self.world, self.device = rank, world, device
        shard = full_prob.shard_points(rank, world)
        active = np.zeros(full_prob.n_cam, dtype=np.uint8)
        active[np.unique(full_prob.obs_cam)] = 1
        self._point_range = shard.meta["point_range"]
        self._template = (np.ascontiguousarray(shard.cam6, np.float64).copy(), np.ascontiguousarray(shard.pt3, np.float64).copy())
        cam6, pt3 = self._template
        oc = np.ascontiguousarray(shard.obs_cam, np.int32)
        op = np.ascontiguousarray(shard.obs_pt, np.int32)
        oxy = np.ascontiguousarray(shard.obs_xy, np.float64)
        self._h = C.c_void_p()
        L = capi.lib()
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        capi._check(L.sfmba_problem_create_ex(
            C.c_int(device), C.c_int(precision), C.c_int(flags), C.c_int(shard.n_cam), cam6.ctypes.data_as(dp),
            active.ctypes.data_as(C.POINTER(C.c_ubyte)), C.c_int(shard.n_pt), pt3.ctypes.data_as(dp), C.c_int64(shard.n_obs),
            oc.ctypes.data_as(ip), op.ctypes.data_as(ip), oxy.ctypes.data_as(dp), C.c_double(shard.focal),
            C.c_int(rank), C.c_int(world), C.byref(self._h)))
        self.L = L
        self.stream = torch.cuda.ExternalStream(L.sfmba_problem_stream(self._h), device=torch.device("cuda", device))
        dev = "cuda:%d" % device
        self.reduce_t = torch.as_tensor(_DevicePtr(L.sfmba_shard_reduce_buf(self._h), L.sfmba_shard_reduce_len(self._h)), device=dev)
        self.setup_t = torch.as_tensor(_DevicePtr(L.sfmba_shard_setup_buf(self._h), L.sfmba_shard_setup_len(self._h)), device=dev)
        self.scalars_t = torch.as_tensor(_DevicePtr(L.sfmba_shard_scalars_buf(self._h), 80), device=dev)

    # -- protocol --
    def begin(self, opt):
        self._opt = opt
        capi._check(self.L.sfmba_shard_begin(self._h, C.byref(opt)))

    def setup_finish(self):
        capi._check(self.L.sfmba_shard_setup_finish(self._h))

    def partial_build(self):
        capi._check(self.L.sfmba_shard_partial_build(self._h))

    def solve_update(self):
        capi._check(self.L.sfmba_shard_solve_update(self._h))

    def finish(self):
        done = C.c_int(0)
        capi._check(self.L.sfmba_shard_finish(self._h, C.byref(done)))
        return bool(done.value)

    def end(self):
        summ = SfmbaSummary()
        capi._check(self.L.sfmba_shard_end(self._h, C.byref(summ)))
        return summ.as_dict()

    def all_reduce(self, dist, which, group=None, n_floats=None):
        t = {"setup": self.setup_t, "reduce": self.reduce_t, "scalars": self.scalars_t}[which]
        if n_floats is not None:                            # the fp32 exchange: the head of the same buffer, viewed as float32
            t = t.view(self.torch.float32)[:n_floats]
        with self.torch.cuda.stream(self.stream):           # the collective is ordered on the solver's own stream
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)

    def get_params(self):
        """(cam6 of all cameras, pt3 of this rank's point range, focal)."""
        cam6, pt3 = self._template[0].copy(), self._template[1].copy()
        focal = C.c_double(0.0)
        dp = C.POINTER(C.c_double)
        capi._check(self.L.sfmba_problem_get_params(self._h, cam6.ctypes.data_as(dp), pt3.ctypes.data_as(dp), C.byref(focal)))
        return cam6, pt3, focal.value

    def reset(self):
        capi._check(self.L.sfmba_problem_reset(self._h))

    def close(self):
        if self._h:
            self.L.sfmba_problem_destroy(self._h)
            self._h = C.c_void_p()


class HipRowShardBackend(HipShardBackend):
    """Rank-local half of the ROW-SHARDED solve (SFMBA_CREATE_ROW_SHARDED, options.shard_distributed_cg = 3): every rank is given the
    WHOLE problem and owns a range of points, a share of the camera-major list and a range of block rows of the reduced matrix; after
    a solve every rank holds the whole solution.  Driven by solve_sharded_native only."""

    CREATE_ROW_SHARDED = 2

    def __init__(self, full_prob, rank, world, device=0, precision=1, flags=0):
        import torch
        self.torch = torch
        self.rank, self.world, self.device = rank, world, device
        self._point_range = (0, full_prob.n_pt)
        self._template = (np.ascontiguousarray(full_prob.cam6, np.float64).copy(), np.ascontiguousarray(full_prob.pt3, np.float64).copy())
        cam6, pt3 = self._template
        oc = np.ascontiguousarray(full_prob.obs_cam, np.int32)
        op = np.ascontiguousarray(full_prob.obs_pt, np.int32)
        oxy = np.ascontiguousarray(full_prob.obs_xy, np.float64)
        self._h = C.c_void_p()
        L = capi.lib()
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        capi._check(L.sfmba_problem_create_ex(
            C.c_int(device), C.c_int(precision), C.c_int(flags | self.CREATE_ROW_SHARDED), C.c_int(full_prob.n_cam), cam6.ctypes.data_as(dp),
            C.cast(None, C.POINTER(C.c_ubyte)), C.c_int(full_prob.n_pt), pt3.ctypes.data_as(dp), C.c_int64(full_prob.n_obs),
            oc.ctypes.data_as(ip), op.ctypes.data_as(ip), oxy.ctypes.data_as(dp), C.c_double(full_prob.focal),
            C.c_int(rank), C.c_int(world), C.byref(self._h)))
        self.L = L
        self.stream = torch.cuda.ExternalStream(L.sfmba_problem_stream(self._h), device=torch.device("cuda", device))
        dev = "cuda:%d" % device
        self.reduce_t = torch.as_tensor(_DevicePtr(L.sfmba_shard_reduce_buf(self._h), L.sfmba_shard_reduce_len(self._h)), device=dev)
        self.setup_t = torch.as_tensor(_DevicePtr(L.sfmba_shard_setup_buf(self._h), L.sfmba_shard_setup_len(self._h)), device=dev)
        self.scalars_t = torch.as_tensor(_DevicePtr(L.sfmba_shard_scalars_buf(self._h), 80), device=dev)


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
REDUCE_SCATTER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p)


class RcclComm:
    """One RCCL communicator per rank, owned by the C library (sfmba_comm_*: ncclAllReduce over xGMI).  Rank 0 draws the unique
    id; `dist` (torch.distributed, any backend) only carries those 128 bytes to the other ranks."""

    def __init__(self, dist, rank, world, device=0):
        import torch
        L = capi.lib()
        ident = (C.c_ubyte * 128)()
        if rank == 0:
            capi._check(L.sfmba_comm_unique_id(ident))
        if world > 1:
            obj = [bytes(ident)]
            dist.broadcast_object_list(obj, src=0)
            ident = (C.c_ubyte * 128).from_buffer_copy(obj[0])
        self._h = C.c_void_p()
        torch.cuda.set_device(device)
        capi._check(L.sfmba_comm_create(ident, C.c_int(rank), C.c_int(world), C.c_int(device), C.byref(self._h)))
        self.L = L

    def size(self):
        """(world, rank) as RCCL itself reports them (ncclCommCount / ncclCommUserRank)."""
        w, r = C.c_int(0), C.c_int(-1)
        capi._check(self.L.sfmba_comm_size(self._h, C.byref(w), C.byref(r)))
        return int(w.value), int(r.value)

    def close(self):
        if self._h:
            self.L.sfmba_comm_destroy(self._h)
            self._h = C.c_void_p()


def solve_sharded_native(backend, opt, comm=None, dist=None, group=None):
    """The same LM loop inside the C library (sfmba_problem_solve_sharded): ONE call per solve, the collectives enqueued on the
    solver's stream by the library itself -- `comm` (RcclComm): ncclAllReduce; otherwise a callback into torch.distributed
    (any backend; used by the gloo tests).  world == 1 needs neither."""
    L = backend.L
    summ = SfmbaSummary()
    keep = None
    fn32 = None
    rs = None
    ag = None
    if comm is not None:
        fn, ctx = C.cast(L.sfmba_comm_allreduce, ALLREDUCE_FN), comm._h
        fn32 = C.cast(L.sfmba_comm_allreduce_f32, ALLREDUCE_FN)
        rs = C.cast(L.sfmba_comm_reduce_scatter, REDUCE_SCATTER_FN)
        ag = C.cast(L.sfmba_comm_allgather, ALLGATHER_FN)
    elif dist is not None and backend.world > 1:
        by_ptr = {int(L.sfmba_shard_setup_buf(backend._h)): "setup", int(L.sfmba_shard_reduce_buf(backend._h)): "reduce",
                  int(L.sfmba_shard_scalars_buf(backend._h)): "scalars"}

        def _cb(_ctx, buf, n, _stream):
            try:
                backend.all_reduce(dist, by_ptr[int(buf)], group)
                return 0
            except Exception:                      # never let an exception cross the C boundary
                return 1
        def _cb32(_ctx, buf, n, _stream):
            try:
                backend.all_reduce(dist, by_ptr[int(buf)], group, n_floats=int(n))
                return 0
            except Exception:
                return 1
        def _any(buf, n, typestr):
            t = backend.torch.as_tensor(_DevicePtrT(buf, n, typestr), device="cuda:%d" % backend.device)
            with backend.torch.cuda.stream(backend.stream):
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)

        def _cb_generic(_ctx, buf, n, _stream):
            try:
                key = int(buf)
                if key in by_ptr:
                    backend.all_reduce(dist, by_ptr[key], group)
                else:                              # the distributed CG's vectors: any device buffer of n doubles
                    _any(buf, int(n), "<f8")
                return 0
            except Exception:
                return 1

        def _rs(_ctx, send, recv, n, is_f32, _stream):
            # gloo has no reduce-scatter: all-reduce the whole send buffer (world chunks of n values); chunk `rank` is then in place
            try:
                _any(send, int(n) * backend.world, "<f4" if is_f32 else "<f8")
                return 0
            except Exception:
                return 1
        def _ag(_ctx, buf, nbytes, _stream):
            # gloo has no all-gather of device tensors: one broadcast per slice (bytes, no arithmetic)
            try:
                t = backend.torch.as_tensor(_DevicePtrT(buf, int(nbytes) * backend.world, "|u1"), device="cuda:%d" % backend.device)
                with backend.torch.cuda.stream(backend.stream):
                    for r in range(backend.world):
                        dist.broadcast(t[r * int(nbytes):(r + 1) * int(nbytes)], src=r, group=group)
                return 0
            except Exception:
                return 1
        fn = ALLREDUCE_FN(_cb_generic)
        fn32 = ALLREDUCE_FN(_cb32)
        rs = REDUCE_SCATTER_FN(_rs)
        ag = ALLGATHER_FN(_ag)
        keep = (fn, fn32, rs, ag)
        ctx = None
    else:
        fn, ctx = C.cast(None, ALLREDUCE_FN), None
    # the single-precision all-reduce is optional (exchange (B) in fp32 where the CG stores the matrix in fp32, include/sfmba.h)
    capi._check(L.sfmba_problem_set_allreduce_f32(backend._h, fn32 if fn32 is not None else C.cast(None, ALLREDUCE_FN)))
    capi._check(L.sfmba_problem_set_reduce_scatter(backend._h, rs if rs is not None else C.cast(None, REDUCE_SCATTER_FN)))
    capi._check(L.sfmba_problem_set_allgather(backend._h, ag if ag is not None else C.cast(None, ALLGATHER_FN)))
    capi._check(L.sfmba_problem_solve_sharded(backend._h, C.byref(opt), fn, ctx, C.byref(summ)))
    del keep
    out = summ.as_dict()
    ex = (C.c_int64 * 4)()
    capi._check(L.sfmba_shard_last_exchange(backend._h, ex))
    out["exchange_bytes"] = [int(ex[0]), int(ex[1]), int(ex[2])]
    out["exchange_b_fp32"] = bool(ex[3] & 1)
    out["distributed_cg"] = bool(ex[3] & 2)
    out["implicit_schur_cg"] = bool(ex[3] & 4)
    out["row_sharded"] = bool(ex[3] & 8)
    return out


def solve_sharded(backend, dist, opt, group=None):
    """The LM loop of one rank.  `dist` is torch.distributed (or any object with the same all_reduce API).
    Exactly three collectives are issued: one before the first iteration (column norms for the Jacobi
    scaling, ||x||), then two per LM iteration (reduced system + linearisation scalars; trial-step scalars)."""
    backend.begin(opt)
    backend.all_reduce(dist, "setup", group)
    backend.setup_finish()
    while True:
        backend.partial_build()
        backend.all_reduce(dist, "reduce", group)
        backend.solve_update()
        backend.all_reduce(dist, "scalars", group)
        if backend.finish():
            break
    return backend.end()
