"""CPU phase backend for the sharded LM choreography (TEST INFRASTRUCTURE: numpy + the oracle's Jet
Jacobians).  It implements the same protocol as sfm_toy_library_amd.sharded.HipShardBackend --
begin / setup_finish / partial_build / solve_update / finish / end / all_reduce -- with dense numpy
algebra, so that sharded.solve_sharded() can be exercised with world_size 2 over gloo on a machine
without GPUs and compared with the single-process oracle."""
import numpy as np
import torch

from oracle import oracle_py as oracle

SCAL = 80


class CpuShardBackend:
    def __init__(self, full_prob, rank, world):
        self.rank, self.world = rank, world
        self.full = full_prob
        self.shard = full_prob.shard_points(rank, world)
        self.active = np.unique(full_prob.obs_cam)
        self.slot = -np.ones(full_prob.n_cam, dtype=int)
        self.slot[self.active] = np.arange(len(self.active))
        self.nc = len(self.active)
        self.d = 6 * self.nc + 1
        self.reduce_t = torch.zeros(self.d * self.d + 3 * self.d + SCAL, dtype=torch.float64)
        self.setup_t = self.reduce_t[self.d * self.d + self.d:]
        self.scalars_t = self.reduce_t[-SCAL:]

    # helpers -----------------------------------------------------------------------------------
    def _jac(self, cam, pt, f):
        s = self.shard
        res, jc, jp, jf = oracle.eval_jacobian(s, cam, pt, f)
        n, npl = s.n_obs, s.n_pt
        J = np.zeros((2 * n, self.d + 3 * npl))
        for k in range(n):
            j, i = self.slot[s.obs_cam[k]], s.obs_pt[k]
            J[2 * k:2 * k + 2, 6 * j:6 * j + 6] = jc[k]
            J[2 * k:2 * k + 2, self.d - 1] = jf[k]
            J[2 * k:2 * k + 2, self.d + 3 * i:self.d + 3 * i + 3] = jp[k]
        return res.ravel(), J

    def _views(self):
        d = self.d
        S = self.reduce_t[:d * d].view(d, d).numpy()
        rhs = self.reduce_t[d * d:d * d + d].numpy()
        udiag = self.reduce_t[d * d + d:d * d + 2 * d].numpy()
        bc = self.reduce_t[d * d + 2 * d:d * d + 3 * d].numpy()
        return S, rhs, udiag, bc, self.scalars_t.numpy()

    # protocol ----------------------------------------------------------------------------------
    def begin(self, opt):
        self.opt = opt
        s = self.shard
        self.cam = s.cam6.copy(); self.pt = s.pt3.copy(); self.f = float(s.focal)
        self.radius = opt.initial_radius; self.dec = 2.0; self.iter = 0; self.term = None; self.msg = ""
        self.invalid = 0
        self.res, self.J = self._jac(self.cam, self.pt, self.f)
        n2 = np.sum(self.J ** 2, axis=0)
        self.reduce_t.zero_()
        S, rhs, udiag, bc, scal = self._views()
        udiag[:] = n2[:self.d]                         # camera + focal column norms (partial)
        self.pt_scale = 1.0 / (1.0 + np.sqrt(n2[self.d:])) if opt.jacobi_scaling else np.ones(3 * s.n_pt)
        w = 1.0 if self.rank == 0 else 0.0
        scal[0] = w * (np.sum(self.cam[self.active] ** 2) + self.f ** 2) + np.sum(self.pt ** 2)
        self.new_lin = True

    def setup_finish(self):
        S, rhs, udiag, bc, scal = self._views()
        self.c_scale = 1.0 / (1.0 + np.sqrt(udiag.copy())) if self.opt.jacobi_scaling else np.ones(self.d)
        self.x_norm = float(np.sqrt(scal[0]))

    def partial_build(self):
        d = self.d
        if self.new_lin and self.iter > 0:
            self.res, self.J = self._jac(self.cam, self.pt, self.f)
        Js = self.J * np.concatenate([self.c_scale, self.pt_scale])
        self.Js = Js
        H = Js.T @ Js
        g = Js.T @ self.res
        Hcc, Hcp, Hpp = H[:d, :d], H[:d, d:], H[d:, d:]
        dp = np.clip(np.diag(Hpp), self.opt.min_lm_diagonal, self.opt.max_lm_diagonal) / self.radius
        self.Vinv = np.linalg.inv(Hpp + np.diag(dp))
        self.Hcp, self.gp = Hcp, g[d:]
        self.reduce_t.zero_()
        S, rhs, udiag, bc, scal = self._views()
        S[:] = Hcc - Hcp @ self.Vinv @ Hcp.T
        rhs[:] = g[:d] - Hcp @ self.Vinv @ g[d:]
        udiag[:] = np.diag(Hcc)
        bc[:] = g[:d]
        scal[0] = float(self.res @ self.res)
        scal[16 + self.rank] = float(np.max(np.abs(g[d:] / self.pt_scale))) if len(self.pt_scale) else 0.0

    def solve_update(self):
        d = self.d
        S, rhs, udiag, bc, scal = self._views()
        if self.iter == 0 and self.new_lin:
            self.cost = 0.5 * scal[0]
            self.initial_cost = self.cost
        if self.new_lin:
            self.gmax = max(np.max(np.abs(bc / self.c_scale)), np.max(scal[16:16 + self.world]))
            self.new_lin = False
        dc = np.clip(udiag, self.opt.min_lm_diagonal, self.opt.max_lm_diagonal) / self.radius
        z = np.linalg.solve(S + np.diag(dc), rhs)
        yp = self.Vinv @ (self.gp - self.Hcp.T @ z)
        step = -np.concatenate([z, yp])
        m = self.Js @ step
        model = -float(m @ (self.res + 0.5 * m))
        dlt_c = step[:d] * self.c_scale
        dlt_p = step[d:] * self.pt_scale
        self.cam_n = self.cam.copy()
        self.cam_n[self.active] = self.cam[self.active] + dlt_c[:d - 1].reshape(-1, 6)
        self.f_n = self.f + dlt_c[d - 1]
        self.pt_n = self.pt + dlt_p.reshape(-1, 3)
        res_n, cost_n = oracle.eval_residuals(self.shard, self.cam_n, self.pt_n, self.f_n)
        w = 1.0 if self.rank == 0 else 0.0
        scal[:] = 0.0
        scal[0] = 2.0 * cost_n
        scal[1] = model
        scal[2] = w * float(dlt_c @ dlt_c) + float(dlt_p @ dlt_p)
        scal[3] = w * (np.sum(self.cam_n[self.active] ** 2) + self.f_n ** 2) + np.sum(self.pt_n ** 2)

    def finish(self):
        """Mirror of k_lm_control (TrustRegionMinimizer accept/reject)."""
        o = self.opt
        S, rhs, udiag, bc, scal = self._views()
        self.iter += 1
        cand, model, step_norm, xnew = 0.5 * scal[0], scal[1], np.sqrt(scal[2]), np.sqrt(scal[3])
        if not (model > 0):
            self.invalid += 1
            if self.invalid >= o.max_consecutive_invalid_steps:
                self.term, self.msg = 2, "invalid steps"
            self.radius *= 0.5
        else:
            self.invalid = 0
            if step_norm <= o.parameter_tolerance * (self.x_norm + o.parameter_tolerance):
                self.term, self.msg = 0, "Parameter tolerance reached."
            elif abs(self.cost - cand) <= o.function_tolerance * self.cost:
                self.term, self.msg = 0, "Function tolerance reached."
            else:
                rho = (self.cost - cand) / model
                if rho > o.min_relative_decrease:
                    self.cam, self.pt, self.f, self.cost, self.x_norm = self.cam_n, self.pt_n, self.f_n, cand, xnew
                    self.radius = min(o.max_radius, self.radius / max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3))
                    self.dec = 2.0
                    self.new_lin = True
                else:
                    self.radius /= self.dec
                    self.dec *= 2.0
        if self.term is None and self.iter >= o.max_iters:
            self.term, self.msg = 1, "Maximum number of iterations reached."
        return self.term is not None

    def end(self):
        return dict(termination=self.term, iterations=self.iter, initial_cost=self.initial_cost, final_cost=self.cost, message=self.msg)

    def all_reduce(self, dist, which, group=None):
        t = {"setup": self.setup_t, "reduce": self.reduce_t, "scalars": self.scalars_t}[which]
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
